from .capi import TRT_LLM_PLUGIN_NAMESPACE
