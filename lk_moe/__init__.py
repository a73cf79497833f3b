"""Drop-in `lk_moe` module for LvLLM (imported at routed_experts.py:37-38 when
LVLLM_MOE_NUMA_ENABLED=1).  The implementation lives in lvllm_amd.lk_moe_api and runs on MI355X
through liblkm.so; see INTEGRATION.md."""
from lvllm_amd.lk_moe_api import *  # noqa: F401,F403
from lvllm_amd.lk_moe_api import __all__  # noqa: F401

__version__ = "2.3.3+mi355x"
