// routing_dev.h -- device side of the router: one 64-lane wavefront per token row (see routing.hip for the
// reference lines and the bit-reproducible arithmetic sequence).  Shared by routing.hip (stand-alone router
// launches) and dispatch.hip (router + scatter metadata in one launch for decode batches).
#pragma once
#include "lkm_common.h"

namespace lkm {

constexpr int kMaxSlots = 8;  // E <= 512: up to 8 register slots per lane

__device__ __forceinline__ float load_logit(const void* p, int dt, size_t i) {
    if (dt == LKM_DT_F32) return ((const float*)p)[i];
    unsigned short h = ((const unsigned short*)p)[i];
    return dt == LKM_DT_BF16 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}

// Where a row's logits come from: a tensor in any of the three dtypes, or -- behind the router GEMM
// (router_gemm.hip) -- n_slabs f32 split-K partials that are summed in ascending slab order, plus the
// gate bias, optionally rounded to the gate's output dtype (F.linear in bf16/f16), optionally copied out.
struct LogitSrc {
    const void* p;
    int dt;
    int n_slabs;
    long long slab_stride;   // floats between slabs
    const float* gate_bias;  // [E] or null
    int round_dt;            // LKM_DT_F32 = keep fp32
    float* logits_out;       // [M,E] fp32 or null
    __device__ __forceinline__ float load(int row, int E, int e) const {
#pragma clang fp contract(off)
        const size_t i = (size_t)row * E + e;
        if (n_slabs <= 1 && !gate_bias && round_dt == LKM_DT_F32 && !logits_out) return load_logit(p, dt, i);
        float v = load_logit(p, dt, i);
        for (int s0 = 1; s0 < n_slabs; s0 += 8) {   // up to 8 independent loads in flight, summed in slab order
            float part[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                part[q] = s0 + q < n_slabs ? ((const float*)p)[(size_t)(s0 + q) * slab_stride + i] : 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) v = v + part[q];
        }
        if (gate_bias) v = v + gate_bias[e];
        if (round_dt == LKM_DT_BF16) v = bf16_bits_to_f32(f32_to_bf16_bits(v));
        if (round_dt == LKM_DT_F16) v = f16_bits_to_f32(f32_to_f16_bits(v));
        if (logits_out) logits_out[i] = v;
        return v;
    }
};

// A row occupies LPR lanes of the wavefront (64, or 16 when E <= 16: four rows per wavefront) and SLOTS register
// slots per lane: expert e lives in slot e / LPR of lane e % LPR of its row's lane group.  Reductions are xor
// butterflies LPR/2 .. 1; lanes / slots beyond E hold the neutral element (-inf, +0), and adding +0 or taking the
// maximum with -inf is exact, so every (SLOTS, LPR) that covers E gives the bits of the 8-slot, 64-lane form.
template <int LPR, typename T>
__device__ __forceinline__ T group_xor(T v, int m) { return __shfl_xor(v, m, 64); }

// Maximum over the 64 lanes by DPP (no LDS crossbar: ~10 cycles a step instead of ~100 for a ds_bpermute butterfly):
// inclusive prefix maximum inside each row of 16 (row_shr 1, 2, 4, 8; lanes without a source keep -inf), then the
// rows' last lanes fan out (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3): lane 63 holds the maximum.
// The maximum is exact in any order, so the bits of the selection do not depend on how it is reduced.
__device__ __forceinline__ float wave_max64(float v) {
    const int ninf = __builtin_bit_cast(int, -__builtin_inff());
#define LKM_DPP_MAX(CTRL, RMASK)                                                                              \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v), CTRL, \
                                                                       RMASK, 0xf, false)))
    LKM_DPP_MAX(0x111, 0xf);
    LKM_DPP_MAX(0x112, 0xf);
    LKM_DPP_MAX(0x114, 0xf);
    LKM_DPP_MAX(0x118, 0xf);
    LKM_DPP_MAX(0x142, 0xa);
    LKM_DPP_MAX(0x143, 0xc);
#undef LKM_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// scores for one row, in registers: sc[s] = score of expert s*LPR+sub (0 for e >= E)
template <int SLOTS, int LPR>
__device__ __forceinline__ void row_scores(const LogitSrc& src, int row, int E, int sub,
                                           int scoring, float (&sc)[SLOTS]) {
#pragma clang fp contract(off)
    float v[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        int e = s * LPR + sub;
        v[s] = (e < E) ? src.load(row, E, e) : -__builtin_inff();
    }
    if (scoring == 0) {
        float mx = v[0];
#pragma unroll
        for (int s = 1; s < SLOTS; ++s) mx = fmaxf(mx, v[s]);
        if constexpr (LPR == 64) {
            mx = wave_max64(mx);             // exact in any order (the SUM below keeps the butterfly: its order is part of the bits)
        } else {
#pragma unroll
            for (int m = LPR / 2; m > 0; m >>= 1) mx = fmaxf(mx, group_xor<LPR>(mx, m));
        }
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            int e = s * LPR + sub;
            if (e < E) {
                v[s] = lkm_expf(v[s] - mx);
                sum += v[s];
            } else {
                v[s] = 0.0f;
            }
        }
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) sum = sum + group_xor<LPR>(sum, m);
        float rinv = 1.0f / sum;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) sc[s] = v[s] * rinv;
    } else {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            int e = s * LPR + sub;
            sc[s] = (e < E) ? 1.0f / (1.0f + lkm_expf(-v[s])) : 0.0f;
        }
    }
}

// group-wide arg-max of (value, index) with "lowest index wins ties"; also carries a payload.
template <int LPR>
__device__ __forceinline__ void wave_argmax(float& bv, int& be, float& bp) {
#pragma unroll
    for (int m = LPR / 2; m > 0; m >>= 1) {
        float ov = group_xor<LPR>(bv, m);
        int oe = group_xor<LPR>(be, m);
        float op = group_xor<LPR>(bp, m);
        if (ov > bv || (ov == bv && oe < be)) {
            bv = ov;
            be = oe;
            bp = op;
        }
    }
}

// One selection round of a row held by a whole wavefront: the largest ch (lowest expert index among equals, like the
// butterfly's tie rule) -> (be, bp = its score), and that entry is struck out.  Wave maximum by DPP, the index by a
// ballot per register slot (slot-major = expert order) and a scalar find-first, the payload by v_readlane: a round is
// ~200 cycles instead of six dependent 3-value shuffles.
template <int SLOTS>
__device__ __forceinline__ void select_max64(float (&ch)[SLOTS], const float (&sc)[SLOTS], int lane, int& be, float& bp) {
    float m = ch[0];
#pragma unroll
    for (int s = 1; s < SLOTS; ++s) m = fmaxf(m, ch[s]);
    const float vmax = wave_max64(m);
    bool found = false;
    be = 0;
    bp = 0.0f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const unsigned long long mk = __ballot(ch[s] == vmax);
        if (!found && mk != 0ull) {                       // wave-uniform
            const int L = __ffsll((long long)mk) - 1;
            be = s * 64 + L;
            bp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc[s]), L));
            if (lane == L) ch[s] = -__builtin_inff();
            found = true;
        }
    }
}

// Plain top-k of one row (softmax / sigmoid scores, optional selection bias).  Lane sub == k of the row's group
// (k < K <= LPR) returns selection k in (w, id): the weight is final (renormalised, scaled).
template <int SLOTS, int LPR>
__device__ __forceinline__ void topk_row(const LogitSrc& src, const float* __restrict__ bias, int row, int E, int K,
                                         int scoring, int renorm, float rsf, int sub, float& w, int& id) {
#pragma clang fp contract(off)
    float sc[SLOTS], ch[SLOTS];
    row_scores<SLOTS, LPR>(src, row, E, sub, scoring, sc);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        int e = s * LPR + sub;
        if (__builtin_isnan(sc[s]) || __builtin_isinf(sc[s])) sc[s] = 0.0f;  // :466-471
        if (e < E)
            ch[s] = bias ? sc[s] + bias[e] : sc[s];
        else
            ch[s] = -__builtin_inff();
    }
    float sel_sum = 0.0f;
    w = 0.0f;
    id = -1;
    for (int k = 0; k < K; ++k) {
        float bp;
        int be;
        if constexpr (LPR == 64) {
            select_max64<SLOTS>(ch, sc, sub, be, bp);
        } else {
            float bv = ch[0];
            bp = sc[0];
            be = sub;
#pragma unroll
            for (int s = 1; s < SLOTS; ++s) {
                if (ch[s] > bv) {
                    bv = ch[s];
                    bp = sc[s];
                    be = s * LPR + sub;
                }
            }
            wave_argmax<LPR>(bv, be, bp);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (be == s * LPR + sub) ch[s] = -__builtin_inff();
        }
        if (sub == k) {
            w = bp;
            id = be;
        }
        if (renorm) sel_sum += bp;       // ascending k, the same value in every lane of the group
    }
    float scale = rsf;
    if (renorm) scale /= (sel_sum > 0.0f ? sel_sum : 1.0f);  // :581-592
    w *= scale;
}

// Group-limited top-k of one row by one whole wavefront; lds_ch = SLOTS * 64 floats of LDS owned by this wavefront.
template <int SLOTS>
__device__ __forceinline__ void grouped_topk_row(const LogitSrc& src, const float* __restrict__ bias, int row, int E,
                                                 int K, int n_group, int topk_group, int scoring, int renorm,
                                                 float rsf, int lane, float* lds_ch, float& w, int& id) {
#pragma clang fp contract(off)
    const int gsz = E / n_group;
    float sc[SLOTS], ch[SLOTS];
    row_scores<SLOTS, 64>(src, row, E, lane, scoring, sc);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        int e = s * 64 + lane;
        ch[s] = (e < E) ? (bias ? sc[s] + bias[e] : sc[s]) : -__builtin_inff();
        lds_ch[s * 64 + lane] = ch[s];
    }
    // the row's scores are written and read by this wavefront only: LDS operations of one wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane g scores group g in ascending expert order (the order of the CPU restatement the tests compare with)
    float gs = -__builtin_inff();
    if (lane < n_group) {
        const float* c = &lds_ch[lane * gsz];
        if (bias) {
            float a = -__builtin_inff(), b = -__builtin_inff();
            for (int i = 0; i < gsz; ++i) {
                float x = c[i];
                if (x > a) {
                    b = a;
                    a = x;
                } else if (x > b) {
                    b = x;
                }
            }
            gs = (gsz > 1) ? a + b : a;
        } else {
            float a = c[0];
            for (int i = 1; i < gsz; ++i) a = (c[i] > a) ? c[i] : a;
            gs = a;
        }
    }
    unsigned long long keep = 0ull;
    bool taken = !(lane < n_group);
    for (int t = 0; t < topk_group; ++t) {
        // lanes already taken / out of range must never win, not even on ties with -inf values: they are left out of
        // the ballot (lowest lane among the equal maxima of the lanes still in play)
        const float vmax = wave_max64(taken ? -__builtin_inff() : gs);
        const unsigned long long mk = __ballot(!taken && gs == vmax);
        const int L = mk != 0ull ? __ffsll((long long)mk) - 1 : 0;
        keep |= 1ull << L;
        if (lane == L) taken = true;
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        int e = s * 64 + lane;
        if (e < E && !((keep >> (e / gsz)) & 1ull)) ch[s] = -__builtin_inff();
    }
    float sum = 0.0f;
    w = 0.0f;
    id = -1;
    for (int k = 0; k < K; ++k) {
        float bp;
        int be;
        select_max64<SLOTS>(ch, sc, lane, be, bp);
        if (lane == k) {
            w = bp;
            id = be;
        }
        sum += bp;
    }
    if (renorm) w = w / sum;
    if (rsf != 1.0f) w = w * rsf;
}

// What the router computes, as one argument block; route_rows() is the body shared by the stand-alone router kernels
// (routing.hip) and the fused router + scatter launch (dispatch.hip).
struct RouteArgs {
    LogitSrc src;
    const float* bias;      // selection bias [E] or null
    int M, E, K;            // E = the ROUTER's expert count (global ids)
    int n_group, topk_group;   // n_group > 0: group-limited top-k
    int scoring, renorm;
    float rsf;
    float* out_w;           // [M][K] fp32
    int32_t* out_ids;       // [M][K] int32 (global ids)
};

// register slots a row needs (1, 2, 4 or 8) and rows per wavefront for a router shape
__host__ __device__ inline int route_slots(int E) { return E <= 64 ? 1 : (E <= 128 ? 2 : (E <= 256 ? 4 : 8)); }
__host__ __device__ inline int route_rows_per_wave(int E, int K, int n_group) { return (n_group <= 0 && E <= 16 && K <= 16) ? 4 : 1; }

// Routes rows  first_row + i * row_step  (i = 0, 1, ...; this wavefront's rows of a pass) -- every wavefront of the
// launch calls it with the same trip count.  lds_ch: SLOTS*64 floats per wavefront (group-limited routing only);
// s_ids: optional LDS copy of the ids [M*K].
template <int SLOTS>
__device__ __forceinline__ void route_rows(const RouteArgs& ra, int first_row, int row_step, int lane, float* lds_ch,
                                           int32_t* s_ids) {
    const bool four = route_rows_per_wave(ra.E, ra.K, ra.n_group) == 4;
    for (int base = first_row; base < ra.M; base += row_step) {
        float w;
        int id, row, sub;
        if (ra.n_group > 0) {
            row = base;
            sub = lane;
            grouped_topk_row<SLOTS>(ra.src, ra.bias, row, ra.E, ra.K, ra.n_group, ra.topk_group, ra.scoring, ra.renorm,
                                    ra.rsf, lane, lds_ch, w, id);
        } else if (SLOTS == 1 && four) {
            row = base + (lane >> 4);
            sub = lane & 15;
            const int rowc = row < ra.M ? row : ra.M - 1;     // a lane group past the last row repeats it, writes nothing
            topk_row<1, 16>(ra.src, ra.bias, rowc, ra.E, ra.K, ra.scoring, ra.renorm, ra.rsf, sub, w, id);
        } else {
            row = base;
            sub = lane;
            topk_row<SLOTS, 64>(ra.src, ra.bias, row, ra.E, ra.K, ra.scoring, ra.renorm, ra.rsf, sub, w, id);
        }
        if (row < ra.M && sub < ra.K) {
            ra.out_w[(size_t)row * ra.K + sub] = w;
            ra.out_ids[(size_t)row * ra.K + sub] = id;
            if (s_ids) s_ids[row * ra.K + sub] = id;
        }
    }
}

}  // namespace lkm
