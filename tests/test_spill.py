"""Host logic of the spill tier (lvllm_amd/spill.py): the prefetch window from the reference's environment
variable and the grouping of the experts that have routed rows.  The device path is tests/test_gpu_spill.py."""
from lvllm_amd.spill import group_plan, prefetch_window_from_env


def test_prefetch_window_follows_the_reference_environment_variable():
    # vllm/envs.py:1942-1943: int(os.getenv("LVLLM_GPU_PREFETCH_WINDOW", "3"))
    assert prefetch_window_from_env({}) == 3
    assert prefetch_window_from_env({"LVLLM_GPU_PREFETCH_WINDOW": "5"}) == 5
    assert prefetch_window_from_env({"LVLLM_GPU_PREFETCH_WINDOW": "0"}) == 1       # (at least the expert being multiplied)
    assert prefetch_window_from_env({"LVLLM_GPU_PREFETCH_WINDOW": "x"}) == 3


def test_groups_hold_only_experts_with_rows_heaviest_first():
    counts = [0, 7, 0, 3, 9, 0, 1, 3]
    assert group_plan(counts, 2) == [[4, 1], [3, 7], [6]]
    assert group_plan(counts, 3) == [[4, 1, 3], [7, 6]]
    assert group_plan([0, 0, 0], 2) == []
    assert group_plan([5], 4) == [[0]]
    flat = [e for g in group_plan(counts, 2) for e in g]
    assert sorted(flat) == [1, 3, 4, 6, 7] and len(set(flat)) == len(flat)
