from .capi import TRT_LLM_PLUGIN_NAMESPACE
from .plugin import _TRT_LLM_PLUGIN_NAMESPACE, ContextFMHAType, PluginConfig, _load_plugin_lib

__all__ = ['_TRT_LLM_PLUGIN_NAMESPACE', 'TRT_LLM_PLUGIN_NAMESPACE', 'ContextFMHAType', 'PluginConfig',
           '_load_plugin_lib']
