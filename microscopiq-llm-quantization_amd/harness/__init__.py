"""WikiText-2 perplexity harness -- the surface of the reference's llm/{llama,opt}.py eval path
(llm/llama.py:176-284, llm/opt.py:131-252) with the quantisation done by the HIP library."""
from .modelutils import DEV, find_layers  # noqa: F401
