// routing.hip -- router kernels for gfx950 (wave64): fused softmax/sigmoid top-k, group-limited
// top-k, global->local expert id map.
//
// What they compute follows the reference (paths relative to the reference tree):
//   topk_softmax   csrc/libtorch_stable/moe/topk_softmax_kernels.cu:408-592
//   grouped_topk   vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py:112-161
//   id map         vllm/model_executor/layers/fused_moe/routed_experts.py:1332-1342
// How: one 64-lane wavefront per token row, expert e lives in lane e%64 (register slot e/64),
// reductions are 64-wide xor butterflies.  The arithmetic sequence (lkm_expf, per-lane sums in
// ascending e then butterfly 32..1, 1/sum, score*rinv) is restated verbatim in
// oracle/lkm_oracle.c so ids AND weights are bit-reproducible; contraction is off for that reason.
#include "lkm_common.h"
#include "lkm_kernels.h"
#include "routing_dev.h"

namespace lkm {

// One launch routes M rows: 4 wavefronts per workgroup, one row per wavefront (four when E <= 16: a row then takes
// 16 lanes), register slots by expert count (routing_dev.h).
template <int SLOTS>
__global__ __launch_bounds__(256) void route_kernel(const RouteArgs ra) {
    __shared__ float lds_ch[4][SLOTS * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rpw = route_rows_per_wave(ra.E, ra.K, ra.n_group);
    route_rows<SLOTS>(ra, (blockIdx.x * 4 + wv) * rpw, gridDim.x * 4 * rpw, lane, lds_ch[wv], nullptr);
}

static int launch_route(hipStream_t st, const RouteArgs& ra) {
    const int rpw = route_rows_per_wave(ra.E, ra.K, ra.n_group);
    const dim3 grid(ceil_div(ra.M, 4 * rpw)), block(256);
    switch (route_slots(ra.E)) {
    case 1: hipLaunchKernelGGL(route_kernel<1>, grid, block, 0, st, ra); break;
    case 2: hipLaunchKernelGGL(route_kernel<2>, grid, block, 0, st, ra); break;
    case 4: hipLaunchKernelGGL(route_kernel<4>, grid, block, 0, st, ra); break;
    default: hipLaunchKernelGGL(route_kernel<8>, grid, block, 0, st, ra); break;
    }
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}

__global__ void map_ids_kernel(const int32_t* __restrict__ ids, int64_t n,
                               const int32_t* __restrict__ map, int E, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t id = ids[i];
    int32_t c = id < 0 ? 0 : (id > E - 1 ? E - 1 : id);
    out[i] = id < 0 ? -1 : map[c];
}

}  // namespace lkm

using namespace lkm;

extern "C" int lkm_topk_softmax(void* stream, const void* logits, int32_t logits_dtype,
                                const float* bias, int32_t M, int32_t E, int32_t K,
                                int32_t scoring, int32_t renormalize, float routed_scaling,
                                float* out_weights, int32_t* out_ids) {
    LKM_REQUIRE(M >= 0 && E > 0 && K > 0 && K <= E && K <= 64, "topk_softmax: bad shape M=%d E=%d K=%d", M, E, K);
    LKM_REQUIRE(E <= kMaxSlots * 64, "topk_softmax: E=%d > %d unsupported", E, kMaxSlots * 64);
    LKM_REQUIRE(logits_dtype >= LKM_DT_F32 && logits_dtype <= LKM_DT_F16, "topk_softmax: bad dtype");
    LKM_REQUIRE(scoring == 0 || scoring == 1, "topk_softmax: scoring must be 0 (softmax) or 1 (sigmoid)");
    if (M == 0) return LKM_OK;
    const LogitSrc src{logits, logits_dtype, 1, 0, nullptr, LKM_DT_F32, nullptr};
    const RouteArgs ra{src, bias, M, E, K, 0, 0, scoring, renormalize, routed_scaling, out_weights, out_ids};
    return launch_route((hipStream_t)stream, ra);
}

extern "C" int lkm_grouped_topk(void* stream, const void* logits, int32_t logits_dtype,
                                const float* bias, int32_t M, int32_t E, int32_t K,
                                int32_t n_group, int32_t topk_group, int32_t scoring,
                                int32_t renormalize, float routed_scaling, float* out_weights,
                                int32_t* out_ids) {
    LKM_REQUIRE(M >= 0 && E > 0 && K > 0 && K <= 64, "grouped_topk: bad shape");
    LKM_REQUIRE(E <= kMaxSlots * 64, "grouped_topk: E=%d > %d unsupported", E, kMaxSlots * 64);
    LKM_REQUIRE(n_group > 0 && n_group <= 64 && E % n_group == 0, "grouped_topk: n_group=%d must divide E=%d and be <= 64", n_group, E);
    LKM_REQUIRE(topk_group > 0 && topk_group <= n_group, "grouped_topk: bad topk_group=%d", topk_group);
    LKM_REQUIRE(K <= topk_group * (E / n_group), "grouped_topk: K=%d exceeds kept experts", K);
    LKM_REQUIRE(logits_dtype >= LKM_DT_F32 && logits_dtype <= LKM_DT_F16, "grouped_topk: bad dtype");
    LKM_REQUIRE(scoring == 0 || scoring == 1, "grouped_topk: scoring must be 0 or 1");
    if (M == 0) return LKM_OK;
    const LogitSrc src{logits, logits_dtype, 1, 0, nullptr, LKM_DT_F32, nullptr};
    const RouteArgs ra{src, bias, M, E, K, n_group, topk_group, scoring, renormalize, routed_scaling, out_weights, out_ids};
    return launch_route((hipStream_t)stream, ra);
}


// ------------------------------------------------------------------ router GEMM
// logits[M,E] = x[M,H] . Wg[E,H]^T, fp32 accumulate: the gate projection the reference runs as F.linear /
// its specialised small-M router GEMMs (moe_runner.py:903-908, router/gate_linear.py:17-34;
// tests/kernels/test_fp32_router_gemm.py:35-37 is the numerical reference: F.linear in fp32).
// The problem is tiny and latency-bound (DSv3: 3.7 MB of gate weights, M <= a few hundred), so it is
// cut into (16-expert tile) x (K slice) x (16-token block) work items, one wavefront each, spread over
// the whole chip; every item writes its f32 partial tile to slab[kslice] and the top-k kernel that
// follows sums the slabs in ascending order while it loads the row (LogitSrc) -- deterministic, no
// atomics, no third launch.  Gate weights = MFMA A operand straight from their [E,H] row-major home
// (16-byte loads), tokens = B operand.
//   16-bit gate weights (dtype of x): v_mfma_f32_16x16x32_{bf16,f16}
//   fp32 gate weights (force_fp32_compute): v_mfma_f32_16x16x4_f32, x widened exactly; one 16-byte
//   weight load feeds 4 MFMAs (MFMA s takes element s of every lane's float4: any k assignment that is
//   the same for A and B is a valid dot product).
constexpr int kRouterSteps = 8;   // MFMA k-steps per wave: the whole K sub-slice is loaded up front

template <int XDT, bool W32, int NT>
__global__ __launch_bounds__(256) void router_gemm_kernel(const unsigned short* __restrict__ x,
                                                          const void* __restrict__ w,
                                                          float* __restrict__ partial, int M, int H, int E,
                                                          int ETG, int KG, int kslice) {
    // workgroup = (group of NT expert tiles, token block, K group); its 4 waves take consecutive K
    // sub-slices of `kslice` elements and are reduced through LDS in wave order -> slab[K group]
    constexpr int Q = W32 ? 16 : 32;                 // k elements one wave-wide MFMA step covers
    __shared__ f32x4 red[3][NT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int tg = blockIdx.x % ETG, kg = (blockIdx.x / ETG) % KG, mb = blockIdx.x / (ETG * KG);
    const int k0 = (kg * 4 + wave) * kslice;
    const int nsteps = k0 < H ? min(kslice, H - k0) / Q : 0;      // wave-uniform
    const int trow = mb * 16 + i;
    const size_t xrow = (size_t)(trow < M ? trow : M - 1) * H;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!W32) {
        u32x4 a[NT][kRouterSteps], b[kRouterSteps];
#pragma unroll
        for (int s = 0; s < kRouterSteps; ++s) {
            const int k = k0 + s * 32 + g * 8;
            const bool on = s < nsteps;
            b[s] = on ? *(const u32x4*)(x + xrow + k) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int erow = (tg * NT + t) * 16 + i;
                a[t][s] = (on && erow < E) ? *(const u32x4*)((const unsigned short*)w + (size_t)erow * H + k)
                                           : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int s = 0; s < kRouterSteps; ++s)
            if (s < nsteps) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = ActT<XDT>::mfma(a[t][s], b[s], acc[t]);
            }
    } else {
        f32x4 a[NT][kRouterSteps];
        u32x2 b[kRouterSteps];
#pragma unroll
        for (int s = 0; s < kRouterSteps; ++s) {
            const int k = k0 + s * 16 + g * 4;
            const bool on = s < nsteps;
            b[s] = on ? *(const u32x2*)(x + xrow + k) : u32x2{0u, 0u};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int erow = (tg * NT + t) * 16 + i;
                a[t][s] = (on && erow < E) ? *(const f32x4*)((const float*)w + (size_t)erow * H + k)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int s = 0; s < kRouterSteps; ++s)
            if (s < nsteps) {
                float xf[4];
                xf[0] = ActT<XDT>::to_f32((unsigned short)(b[s].x & 0xffffu));
                xf[1] = ActT<XDT>::to_f32((unsigned short)(b[s].x >> 16));
                xf[2] = ActT<XDT>::to_f32((unsigned short)(b[s].y & 0xffffu));
                xf[3] = ActT<XDT>::to_f32((unsigned short)(b[s].y >> 16));
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s][q], xf[q], acc[t], 0, 0, 0);
            }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave != 0) return;
    // D layout: lane (g, j=i): experts tile*16 + g*4 + r, token mb*16 + j
    const int tok = mb * 16 + i;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 v = acc[t];
#pragma unroll
        for (int wv = 0; wv < 3; ++wv) v = v + red[wv][t][lane];
        const int e0 = (tg * NT + t) * 16 + g * 4;
        if (tok < M) {
            float* o = partial + (size_t)kg * M * E + (size_t)tok * E + e0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (e0 + r < E) o[r] = v[r];
        }
    }
}

struct RouterPlan {
    int NT, ETG, KG, kslice, MB;
};
static RouterPlan router_plan(int M, int H, int E, bool w32) {
    RouterPlan p;
    const int q = w32 ? 16 : 32;
    const int ET = ceil_div(E, 16);
    p.MB = ceil_div(M, 16);
    p.NT = (p.MB >= 64 && ET >= 4) ? 4 : 1;           // many token blocks: reuse the token fragments
    p.ETG = ceil_div(ET, p.NT);
    // K sub-slice per wave: as short as it takes to put >= 256 workgroups on the chip, at most
    // kRouterSteps MFMA steps, and never more than 8 slabs for the routing kernel to sum
    int want_kg = ceil_div(256, p.ETG * p.MB);
    if (want_kg < 1) want_kg = 1;
    if (want_kg > 8) want_kg = 8;
    int ks = ceil_div(ceil_div(H, 4 * want_kg), q) * q;
    const int max_ks = kRouterSteps * q;
    if (ks > max_ks) ks = max_ks;
    if (ks < q) ks = q;
    p.kslice = ks;
    p.KG = ceil_div(H, 4 * ks);
    return p;
}

// ---- prefill-size router GEMM: the gate projection is ONE more grouped GEMM -- a single "expert" whose
// weight rows are the E router rows and whose token rows are all M tokens -- so at M >= kRouterBigM it runs
// on gemm_tiled_kernel (GEMM2 flavour: tokens staged through LDS once per 128/256-row tile, fp32 split-K
// slabs), which is exactly the slab format the routing kernels sum.  The work items of router_gemm_kernel
// are one MFMA fragment deep and re-read both operands through L2 (655 MB at M=8192; 98 us against 45 us for
// a library GEMM); this path reads x once.  16-bit gate weights only (the tiled kernels take the activation
// dtype); fp32 gate weights keep router_gemm_kernel.
static const int kRouterBigM = 1024;
struct RouterBigPlan {
    int tile_rows, waves, T, U, n_tiles, SK;
    size_t slab_bytes, w_bytes, meta_ints;
    size_t total() const { return slab_bytes + w_bytes + meta_ints * 4; }
};
static RouterBigPlan router_big_plan(int M, int H, int E) {
    RouterBigPlan b;
    b.T = ceil_div(E, 16);
    b.waves = b.T >= 5 ? 8 : 4;
    b.tile_rows = b.waves == 8 ? 256 : 128;
    b.U = ceil_div(H, 64);
    b.n_tiles = ceil_div(M, b.tile_rows);
    const long long wg = (long long)b.n_tiles * ceil_div(b.T, b.waves);
    b.SK = 1;
    while (b.SK < 8 && wg * b.SK < 256 && b.U / (b.SK * 2) >= 4) b.SK *= 2;
    b.slab_bytes = ((size_t)b.SK * M * E * 4 + 255) / 256 * 256;
    b.w_bytes = (size_t)b.T * b.U * 2 * 64 * 16;
    b.meta_ints = 16 + 2 * (size_t)b.n_tiles;
    return b;
}
static bool router_big_ok(int M, int H, int E, bool w32) { return !w32 && M >= kRouterBigM && H % 64 == 0; }

__global__ void router_meta_kernel(int32_t* meta, int M, int rows, int n_tiles) {
    // layout: [0..3] meta (meta[3] = work items), [4] counts, [5..6] offsets, [16..] tile_e, then tile_r0
    for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) {
        meta[16 + i] = 0;
        meta[16 + n_tiles + i] = i * rows;
    }
    if (threadIdx.x == 0) {
        meta[0] = meta[1] = meta[2] = 0;
        meta[3] = n_tiles;
        meta[4] = M;
        meta[5] = 0;
        meta[6] = M;
    }
}

extern "C" int64_t lkm_router_workspace_bytes(int32_t M, int32_t H, int32_t E) {
    if (M <= 0 || H <= 0 || E <= 0) return 0;
    const RouterPlan a = router_plan(M, H, E, false), b = router_plan(M, H, E, true);
    const int64_t small = (int64_t)(a.KG > b.KG ? a.KG : b.KG) * M * E * 4;
    const int64_t big = router_big_ok(M, H, E, false) ? (int64_t)router_big_plan(M, H, E).total() : 0;
    return small > big ? small : big;
}

extern "C" int lkm_router_gemm_topk(void* stream, const void* x, int32_t x_dtype, const void* gate_w,
                                    int32_t w_dtype, const float* gate_bias, const float* score_bias,
                                    int32_t M, int32_t H, int32_t E, int32_t K, int32_t scoring,
                                    int32_t renormalize, float routed_scaling, int32_t n_group,
                                    int32_t topk_group, int32_t logits_dtype, void* workspace,
                                    int64_t workspace_bytes, float* logits_out, float* out_weights,
                                    int32_t* out_ids) {
    LKM_REQUIRE(M >= 0 && E > 0 && K > 0 && K <= E && K <= 64 && H > 0, "router: bad shape M=%d H=%d E=%d K=%d", M, H, E, K);
    LKM_REQUIRE(E <= kMaxSlots * 64, "router: E=%d > %d unsupported", E, kMaxSlots * 64);
    LKM_REQUIRE(x_dtype == LKM_DT_BF16 || x_dtype == LKM_DT_F16, "router: hidden states must be bf16 or fp16");
    LKM_REQUIRE(w_dtype == x_dtype || w_dtype == LKM_DT_F32, "router: gate weights must have the activation dtype or be fp32 (got %d)", w_dtype);
    const bool w32 = w_dtype == LKM_DT_F32;
    LKM_REQUIRE(H % (w32 ? 16 : 32) == 0, "router: hidden_size=%d must be a multiple of %d", H, w32 ? 16 : 32);
    LKM_REQUIRE(logits_dtype == LKM_DT_F32 || logits_dtype == x_dtype, "router: logits_dtype must be fp32 or the activation dtype");
    LKM_REQUIRE(scoring == 0 || scoring == 1, "router: scoring must be 0 (softmax) or 1 (sigmoid)");
    if (n_group > 0) {
        LKM_REQUIRE(n_group <= 64 && E % n_group == 0, "router: n_group=%d must divide E=%d and be <= 64", n_group, E);
        LKM_REQUIRE(topk_group > 0 && topk_group <= n_group, "router: bad topk_group=%d", topk_group);
        LKM_REQUIRE(K <= topk_group * (E / n_group), "router: K=%d exceeds kept experts", K);
    }
    if (M == 0) return LKM_OK;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    int KS;
    if (router_big_ok(M, H, E, w32)) {
        const RouterBigPlan bp = router_big_plan(M, H, E);
        LKM_REQUIRE(workspace && workspace_bytes >= (int64_t)bp.total(), "router: workspace too small (%lld < %lld bytes; lkm_router_workspace_bytes)", (long long)workspace_bytes, (long long)bp.total());
        void* wrep = (char*)workspace + bp.slab_bytes;
        int32_t* meta = (int32_t*)((char*)workspace + bp.slab_bytes + bp.w_bytes);
        const int wf = x_dtype == LKM_DT_BF16 ? LKM_W_BF16 : LKM_W_F16;
        const RepackDims rd{1, E, 1, 0, H, bp.T, bp.U, 0};
        int rc = launch_repack_w(st, wf, gate_w, wrep, rd);
        if (rc != LKM_OK) return rc;
        hipLaunchKernelGGL(router_meta_kernel, dim3(1), dim3(256), 0, st, meta, M, bp.tile_rows, bp.n_tiles);
        LKM_HIP_CHECK(hipGetLastError());
        GemmParams gp{};
        gp.w = wrep;
        gp.T_half = bp.T;
        gp.halves = 1;
        gp.U = bp.U;
        set_w_layout(gp, bp.T, bp.U, wf_loads(wf));
        gp.Kreal = H;
        gp.n_real = E;
        gp.x = x;
        gp.ldx = H;
        gp.x_rows = M;
        gp.top_k = 1;
        gp.rcp_top_k = 1.0f;
        gp.meta = meta;
        gp.counts = meta + 4;
        gp.offsets = meta + 5;
        gp.tile_e = meta + 16;
        gp.tile_r0 = meta + 16 + bp.n_tiles;
        gp.out = part;
        gp.ldo = E;
        gp.sk_stride = (size_t)M * E;
        gp.SK = bp.SK;
        const LaunchCfg cfg{1, bp.tile_rows / 16, 1, bp.SK, bp.tile_rows, bp.waves, 2, 0};
        rc = launch_gemm2_tiled(st, wf, x_dtype, cfg, gp, bp.n_tiles);
        if (rc != LKM_OK) return rc;
        KS = bp.SK;
    } else {
    const RouterPlan pl = router_plan(M, H, E, w32);
    KS = pl.KG;
    LKM_REQUIRE(workspace && workspace_bytes >= (int64_t)KS * M * E * 4, "router: workspace too small (%lld < %lld bytes; lkm_router_workspace_bytes)", (long long)workspace_bytes, (long long)KS * M * E * 4);
    dim3 grid(pl.ETG * pl.KG * pl.MB), block(256);
    const unsigned short* xp = (const unsigned short*)x;
#define LKM_ROUTER_LAUNCH(XDT, W32, NT) \
    hipLaunchKernelGGL((router_gemm_kernel<XDT, W32, NT>), grid, block, 0, st, xp, gate_w, part, M, H, E, pl.ETG, pl.KG, pl.kslice)
    if (x_dtype == LKM_DT_BF16) {
        if (w32) { if (pl.NT == 4) LKM_ROUTER_LAUNCH(LKM_DT_BF16, true, 4); else LKM_ROUTER_LAUNCH(LKM_DT_BF16, true, 1); }
        else     { if (pl.NT == 4) LKM_ROUTER_LAUNCH(LKM_DT_BF16, false, 4); else LKM_ROUTER_LAUNCH(LKM_DT_BF16, false, 1); }
    } else {
        if (w32) { if (pl.NT == 4) LKM_ROUTER_LAUNCH(LKM_DT_F16, true, 4); else LKM_ROUTER_LAUNCH(LKM_DT_F16, true, 1); }
        else     { if (pl.NT == 4) LKM_ROUTER_LAUNCH(LKM_DT_F16, false, 4); else LKM_ROUTER_LAUNCH(LKM_DT_F16, false, 1); }
    }
#undef LKM_ROUTER_LAUNCH
    LKM_HIP_CHECK(hipGetLastError());
    }
    const LogitSrc src{part, LKM_DT_F32, KS, (long long)M * E, gate_bias, logits_dtype, logits_out};
    const RouteArgs ra{src, score_bias, M, E, K, n_group > 0 ? n_group : 0, topk_group, scoring, renormalize, routed_scaling,
                       out_weights, out_ids};
    return launch_route(st, ra);
}

extern "C" int lkm_map_expert_ids(void* stream, const int32_t* ids, int64_t n,
                                  const int32_t* expert_map, int32_t E, int32_t* out) {
    LKM_REQUIRE(n >= 0 && E > 0, "map_expert_ids: bad sizes");
    if (n == 0) return LKM_OK;
    hipLaunchKernelGGL(map_ids_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, ids, n, expert_map, E, out);
    LKM_HIP_CHECK(hipGetLastError());
    return LKM_OK;
}
