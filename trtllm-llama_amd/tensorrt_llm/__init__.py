"""tensorrt_llm-shaped front-end of the MI355X LLaMA decoder path (package version follows T/setup.py:22)."""
__version__ = '0.1.3'
