"""Run by tests/test_gpu_ep.py in a subprocess (needs a GPU): the expert-parallel step over a ONE-rank RCCL group.

Checks, each printed as "<tag> OK":
  a2a-f32 eager   fixed-capacity all-to-all, fp32 return leg: bit-identical to the engine called directly
  a2a-bf16 eager  default return leg (activation dtype): within one bf16 rounding of it
  ar eager        reference-compatible mode: all_gather -> local experts -> reduce_scatter (moe_runner.py:494,
                  communication_op.py:12-14), bit-identical on one rank
  a2a captured    the a2a step recorded in a hipGraph AFTER the communicator exists, replayed on new inputs
  layer ar        RoutedExpertsLayer with an expert_map + the "ar" reduction of its output
  reference FusedMoEKernel   the reference's own modular driver (oracle/_ref/modular_kernel_glue.py) over the bound classes
Exits through os._exit: tearing down a communicator that a live graph still references has hung before."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from lvllm_amd import ops
    from lvllm_amd.ep import ExpertParallelExperts
    from lvllm_amd.layer import RoutedExpertsLayer, RoutingConfig

    E, K, H, I, M = 8, 2, 512, 256, 24
    g = torch.Generator().manual_seed(3)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 8).to(torch.bfloat16).to(dev)
    w2 = (torch.randn((E, H, I), generator=g) / 8).to(torch.bfloat16).to(dev)
    eng = ops.RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16)
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16).to(dev)
    logits = torch.randn((M, E), generator=g).to(dev)
    tw, ids = ops.topk_softmax(logits, K, True)
    want = eng.decode(x, tw, ids).clone()

    def local(rows, lids, ws, dt):
        return eng.forward_rows(rows, ws, lids, out_dtype=dt)

    ep32 = ExpertParallelExperts(local, E, H, mode="a2a", return_dtype=torch.float32)
    got = ep32.forward(x, tw, ids, force_collectives=True)
    assert torch.equal(got, want), float((got - want).abs().max())
    print("a2a-f32 eager OK", flush=True)

    ep16 = ExpertParallelExperts(local, E, H, mode="a2a")
    got = ep16.forward(x, tw, ids, force_collectives=True)
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 2.0 ** -8, err
    assert ep16.overflow_count() == 0
    print(f"a2a-bf16 eager OK (max rel {err:.2e})", flush=True)

    epar = ExpertParallelExperts(local, E, H, mode="ar", return_dtype=torch.float32)
    got = epar.forward(x, tw, ids, force_collectives=True)
    assert torch.equal(got, want)
    # default: the partial sums are reduced in the activation dtype (what the reference all-reduces, moe_runner.py:494)
    got16 = ExpertParallelExperts(local, E, H, mode="ar").forward(x, tw, ids, force_collectives=True,
                                                                  out_dtype=torch.bfloat16)
    assert got16.dtype == torch.bfloat16 and torch.equal(got16, eng.prefill(x, tw, ids))
    print("ar eager OK", flush=True)

    # ---- capture: router top-k + pack + all-to-all + grouped GEMMs + all-to-all + combine in ONE graph
    out = torch.empty((M, H), dtype=torch.float32, device=dev)

    def step():
        tw_, ids_ = ops.topk_softmax(logits, K, True)
        out.copy_(ep32.forward(x, tw_, ids_, force_collectives=True))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    # thread_local: c10d's watchdog thread queries events while this thread captures; in the default global mode a call
    # from any thread invalidates the capture (hipErrorStreamCaptureInvalidated, about one run in three inside the suite)
    with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
        step()
    x2 = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16).to(dev)
    l2 = torch.randn((M, E), generator=g).to(dev)
    x.copy_(x2)
    logits.copy_(l2)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    tw2, ids2 = ops.topk_softmax(l2, K, True)
    want2 = eng.decode(x2, tw2, ids2)
    assert torch.equal(out, want2), float((out - want2).abs().max())
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want2)
    print("a2a captured OK", flush=True)

    # ---- the layer's output before the reduction + the reference-compatible reduction
    _, emap = ops.determine_expert_map(1, 0, E)                  # one rank: no map
    layer = RoutedExpertsLayer(eng, RoutingConfig(K, E), expert_map=emap)
    part = layer.forward(x2, l2).float()                         # [M,H] partial of this rank
    red = torch.empty_like(part)
    dist.reduce_scatter_tensor(red, part.contiguous())           # RS (+ AG = the all-reduce of moe_runner.py:494)
    full = torch.empty_like(part)
    dist.all_gather_into_tensor(full, red)
    torch.cuda.synchronize()
    assert torch.equal(full, part)
    np.testing.assert_allclose(full.cpu().numpy(), want2.cpu().numpy(), atol=2.0 ** -7 * float(want2.abs().max()), rtol=0)
    print("layer ar OK", flush=True)

    # ---- the reference's modular call sequence (modular_kernel.py:1219-1420): prepare -> experts.apply -> finalize
    from lvllm_amd.modular import LkmExperts, LkmPrepareAndFinalize
    pf, ex = LkmPrepareAndFinalize(E, H), LkmExperts()
    a1q, a1q_scale, meta, ids_d, w_d = pf.prepare(x2, tw2, ids2, E, None, False, None, True)
    assert a1q_scale is None and meta is None and ids_d.shape == (M, K) and not a1q.is_contiguous()  # rows are views into the receive buffer
    fused = torch.empty((a1q.size(0), H), dtype=torch.float32, device=dev)
    ex.apply(fused, a1q, w13, w2, w_d, ids_d, "silu", E, None, None, None, None, None, None, False)
    mout = torch.empty((M, H), dtype=torch.float32, device=dev)
    pf.finalize(mout, fused, tw2, ids2, False, ex.finalize_weight_and_reduce_impl())
    torch.cuda.synchronize()
    assert torch.equal(mout, want2), float((mout - want2).abs().max())
    print("modular prepare/apply/finalize OK", flush=True)

    # ---- two micro-batches in flight: the return exchange of batch 0 on the communicator stream under the GEMMs of batch 1
    # (lvllm_amd/ep.py forward_two_microbatches), eager and captured, against the engine called directly
    from lvllm_amd.ep import forward_two_microbatches
    epa = ExpertParallelExperts(local, E, H, mode="a2a", return_dtype=torch.float32, pool_tag="mb0")
    epb = ExpertParallelExperts(local, E, H, mode="a2a", return_dtype=torch.float32, pool_tag="mb1")
    xb = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16).to(dev)
    lb = torch.randn((M, E), generator=g).to(dev)
    twb, idsb = ops.topk_softmax(lb, K, True)
    wantb = eng.decode(xb, twb, idsb).clone()
    side = torch.cuda.Stream()
    o0, o1 = forward_two_microbatches(epa, epb, (x2, tw2, ids2), (xb, twb, idsb), comm_stream=side)
    torch.cuda.synchronize()
    assert torch.equal(o0, want2) and torch.equal(o1, wantb)
    so0, so1 = torch.empty_like(o0), torch.empty_like(o1)

    def step2():
        a_, b_ = forward_two_microbatches(epa, epb, (x2, tw2, ids2), (xb, twb, idsb), comm_stream=side)
        so0.copy_(a_)
        so1.copy_(b_)
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        step2()
    torch.cuda.current_stream().wait_stream(cs)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=cs, capture_error_mode="thread_local"):
        step2()
    so0.zero_(), so1.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(so0, want2) and torch.equal(so1, wantb)
    print("two micro-batches overlapped OK", flush=True)

    # ---- the same sequence owned by the REFERENCE's FusedMoEKernel (modular_kernel.py:1096-1525, 1588-1726; cut out of the
    # reference tree by oracle/make_ref_glue.py) over the real one-rank RCCL group: bound classes, workspace allocation,
    # prepare -> apply -> finalize, the output in the activation dtype
    glue = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "modular_kernel_glue.py")
    if os.path.exists(glue):
        import importlib.util
        import types
        spec = importlib.util.spec_from_file_location("modular_kernel_glue", glue)
        mk = importlib.util.module_from_spec(spec)
        sys.modules["modular_kernel_glue"] = mk
        spec.loader.exec_module(mk)
        from lvllm_amd import modular
        Experts, PF = modular.bind_vllm_base(mk), modular.bind_vllm_prepare_finalize(mk)

        class QC:
            a1_scale = a2_scale = w1_scale = w2_scale = block_shape = quant_dtype = None
        cfg = types.SimpleNamespace(moe_parallel_config=types.SimpleNamespace(dp_size=1, use_ep=True))
        kern = mk.FusedMoEKernel(PF(E, H, pool_tag="refkernel"), Experts(cfg, QC()))
        kout = kern.apply(x2, w13, w2, tw2, ids2, mk.MoEActivation.SILU, E, None, False)
        torch.cuda.synchronize()
        assert kout.dtype == torch.bfloat16 and torch.equal(kout, want2.to(torch.bfloat16)), float((kout.float() - want2).abs().max())
        print("reference FusedMoEKernel OK", flush=True)
    else:
        print("reference FusedMoEKernel SKIPPED (oracle/_ref/modular_kernel_glue.py not built)", flush=True)
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
