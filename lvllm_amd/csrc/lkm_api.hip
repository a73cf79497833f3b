// lkm_api.hip -- host side of liblkm.so: the C ABI declared in include/lkm.h.
//
// One LkmEngine == one lk_moe.MOE_* instance of the reference == the routed experts of one MoE
// layer on one GPU (routed_experts.py:1399-1418).  The engine owns the pre-shuffled weights in
// HBM; scratch (sort metadata, intermediate, split-K partials) is a per-device arena shared by all
// engines because the layers of a model execute one after another on the worker's stream.
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include <map>

#include "lkm_kernels.h"
#include "routing_dev.h"
#include "../../include/lkm_eplb.h"

namespace lkm {

// ------------------------------------------------------------------ error string
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------ which GEMM kernels a step launched
// run_chunk marks the phase (1 = GEMM1, 2 = GEMM2) around its launches; every launcher notes its kernel's host stub
// (LKM_LAUNCH_GEMM, lkm_kernels.h).  Two entries per phase: a hybrid plan launches the streamer and the tile kernel.
static thread_local int g_note_phase = 0;
static thread_local const void* g_note_fn[3][2] = {};
void note_gemm_launch(const void* host_fn) {
    const void** slot = g_note_fn[g_note_phase];
    if (slot[0] == host_fn || slot[1] == host_fn) return;
    if (!slot[0]) slot[0] = host_fn;
    else if (!slot[1]) slot[1] = host_fn;
}
static void note_phase(int phase) {
    g_note_phase = phase;
    if (phase) g_note_fn[phase][0] = g_note_fn[phase][1] = nullptr;
}

// ------------------------------------------------------------------ per-device scratch arena
// Grow-only: blocks that were handed out stay valid until process exit, so hipGraphs captured with
// an older (smaller) arena keep working after a later engine asked for more.
struct Arena {
    int device = -1;
    size_t cap_slots = 0, act_elems = 0, y_elems = 0;
    int E_cap = 0;
    int32_t *counts = nullptr, *offsets = nullptr, *active = nullptr, *meta = nullptr;
    int32_t *sorted_slot = nullptr, *pos_of_slot = nullptr;
    int32_t *tile_e = nullptr, *tile_r0 = nullptr;   // [cap_slots/32 + E_cap + 8] each
    size_t tile_cap = 0;
    int32_t* hist = nullptr;                         // multi-workgroup sort: [chunks][E]
    size_t hist_cap = 0;
    int32_t* items = nullptr;                        // fp8 x fp8 prefill kernel: the two GEMMs' item lists (launch_build_items)
    size_t items_cap = 0;
    unsigned char *xq = nullptr, *aq = nullptr;   // W8A8: fp8 activations (tokens / intermediate rows)
    float *xqs = nullptr, *aqs = nullptr;         //       and their 1x128 scales
    size_t xq_n = 0, aq_n = 0, xqs_n = 0, aqs_n = 0;
    void* act = nullptr;  // [cap_slots][ld_act] 16-bit
    float* y = nullptr;   // split-K partials
    std::vector<void*> retired;
};
static std::mutex g_arena_mu;
static Arena g_arenas[64];

template <typename T>
static int grow(T*& ptr, size_t& have, size_t want, std::vector<void*>& retired) {
    if (want <= have && ptr) return LKM_OK;
    void* np = nullptr;
    hipError_t err = hipMalloc(&np, want * sizeof(T));
    if (err != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(err));
        return LKM_E_NOMEM;
    }
    if (ptr) retired.push_back((void*)ptr);
    ptr = (T*)np;
    have = want;
    return LKM_OK;
}

static int arena_reserve(int device, int E, size_t slots, size_t act_elems, size_t y_elems,
                         size_t xq_n, size_t xqs_n, size_t aq_n, size_t aqs_n, size_t items_n, Arena** out) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    LKM_REQUIRE(device >= 0 && device < 64, "device ordinal %d out of range", device);
    Arena& a = g_arenas[device];
    a.device = device;
    int rc;
    if (E > a.E_cap) {
        size_t h = 0;
        int32_t* p = nullptr;
        // counts[E] offsets[E+1] active[E] meta[kMetaInts] in one block
        size_t n = (size_t)3 * E + 1 + kMetaInts;
        if ((rc = grow(p, h, n, a.retired)) != LKM_OK) return rc;
        if (a.counts) a.retired.push_back(a.counts);
        a.counts = p;
        a.offsets = p + E;
        a.active = a.offsets + E + 1;
        a.meta = a.active + E;
        a.E_cap = E;
    }
    if (slots > a.cap_slots) {
        size_t h1 = 0, h2 = 0;
        int32_t *s1 = nullptr, *s2 = nullptr;
        if ((rc = grow(s1, h1, slots, a.retired)) != LKM_OK) return rc;
        if ((rc = grow(s2, h2, slots, a.retired)) != LKM_OK) return rc;
        if (a.sorted_slot) a.retired.push_back(a.sorted_slot);
        if (a.pos_of_slot) a.retired.push_back(a.pos_of_slot);
        a.sorted_slot = s1;
        a.pos_of_slot = s2;
        a.cap_slots = slots;
    }
    {
        const size_t want = a.cap_slots / 32 + (size_t)a.E_cap + 8;   // smallest token tile: 32 rows
        if (want > a.tile_cap) {
            size_t h1 = 0, h2 = 0;
            int32_t *t1 = nullptr, *t2 = nullptr;
            if ((rc = grow(t1, h1, want, a.retired)) != LKM_OK) return rc;
            if ((rc = grow(t2, h2, want, a.retired)) != LKM_OK) return rc;
            if (a.tile_e) a.retired.push_back(a.tile_e);
            if (a.tile_r0) a.retired.push_back(a.tile_r0);
            a.tile_e = t1;
            a.tile_r0 = t2;
            a.tile_cap = want;
        }
    }
    {
        const size_t want = (a.cap_slots / 1024 + 1) * (size_t)a.E_cap;
        if (a.cap_slots > 4096 && (rc = grow(a.hist, a.hist_cap, want > a.hist_cap ? want : a.hist_cap, a.retired)) != LKM_OK) return rc;
    }
    {
        unsigned short* p = (unsigned short*)a.act;
        if ((rc = grow(p, a.act_elems, act_elems, a.retired)) != LKM_OK) return rc;
        a.act = p;
    }
    if ((rc = grow(a.y, a.y_elems, y_elems, a.retired)) != LKM_OK) return rc;
    if (xq_n && (rc = grow(a.xq, a.xq_n, xq_n, a.retired)) != LKM_OK) return rc;
    if (xqs_n && (rc = grow(a.xqs, a.xqs_n, xqs_n, a.retired)) != LKM_OK) return rc;
    if (aq_n && (rc = grow(a.aq, a.aq_n, aq_n, a.retired)) != LKM_OK) return rc;
    if (aqs_n && (rc = grow(a.aqs, a.aqs_n, aqs_n, a.retired)) != LKM_OK) return rc;
    if (items_n && (rc = grow(a.items, a.items_cap, items_n, a.retired)) != LKM_OK) return rc;
    *out = &a;
    return LKM_OK;
}

}  // namespace lkm

using namespace lkm;

// ------------------------------------------------------------------ engine
struct LkmEngine {
    LkmConfig cfg;
    int device;
    int E, H, I, K;            // local experts, hidden (rounded up to a multiple of 8), intermediate, top_k
    int Hu;                    // hidden size as the caller sees it (== H unless hidden_size % 8 != 0: 16-bit weights only).
                               // The image pads K with zeros anyway; the token rows and the output rows of such a layer
                               // pass through two 16-byte-aligned scratch matrices (pad_x / pad_o, run_device)
    void *pad_x = nullptr, *pad_o = nullptr;
    size_t pad_tokens = 0;
    bool gated, interleaved;
    int wf, adt;
    int wfk;                   // kernel format: wf, LKM_W_FP8_A8 for fp8 W8A8, LKM_W_INT4_PS for the fast int4 mode
    bool a8;
    bool zp;                   // uint4 with zero points (int4_mode ZP): (scale, zero point) pairs in the scale image, kernel format LKM_W_INT4_ZP
    bool ps;                   // int4 fast mode: per-(row, 128-k) activation sums feed the GEMMs (arena xqs / aqs)
    // geometry
    int unitk;
    int T1_half, U1;           // w13: tiles per half, units along H
    int T2, U2;                // w2 : tiles (rows = H), units along I
    int ld_act;                // row stride of the intermediate
    int spu;                   // int4 scales per unit
    // HBM
    void *w13 = nullptr, *w2 = nullptr, *s13 = nullptr, *s2 = nullptr;
    float *gs13 = nullptr, *gs2 = nullptr;   // NVFP4 per-expert multipliers
    int64_t weight_bytes = 0;
    // scratch
    Arena* arena = nullptr;
    size_t cap_tokens = 0;
    // host-IO staging for prefill_host
    void *io_x = nullptr, *io_ids = nullptr, *io_w = nullptr, *io_out = nullptr;
    size_t io_tokens = 0;
    // tuning overrides (<=0 = auto)
    int t_nt1 = 0, t_nt2 = 0, t_kw1 = 0, t_sk2 = 0, t_tb = 0, t_tiled = 0, t_waves = 0, t_hybrid = 0, t_pd1 = 0, t_pd2 = 0, t_xcd = 0, t_pf = 0, t_direct = 0, t_valid_den = 0, t_prof_rep = 0, t_dbg = 0, t_fuse = 0, t_tiled2 = 0, t_fuseq = 0, t_ydt = 0;
    int loads = 2;            // 16-byte loads per lane per (tile, unit)
    // first-call micro-autotune (tuning key "autotune", off by default: the thresholds of pick_cfg stay the plan and
    // results stay reproducible run to run).  When on, the first EAGER call of a decode-sized step shape times two to
    // five candidate plans on the caller's own inputs and remembers the fastest for that shape; later calls -- and the
    // captures that follow the warm-up steps -- take it.
    int t_autotune = 0;
    int t_mixed = 0;          // mixed tile heights (opt-in): n > 0 = experts with more than n rows take 128-row tiles
    struct TunedPlan {
        int pf, tiled, pd1, pd2;
        float us;             // its time when chosen
        char what[96];
        int index = 0;        // position in tune_candidates() of the shape (what an expert-parallel group agrees on)
    };
    std::map<uint64_t, TunedPlan> tuned;
    struct TuneShape { int M, K; };
    std::map<uint64_t, TuneShape> tuned_shape;     // key -> (M, K), to rebuild the candidate list of a remembered plan
    hipEvent_t tune_ev[2] = {};
    // profiling
    bool prof = false;
    hipEvent_t ev[LKM_PROF_N + 1] = {};
    int prof_rep_used = 1;
    hipStream_t prof_stream = nullptr;
    bool prof_valid = false;
    char last_desc[512] = "";
    const void* last_fn[3][2] = {};     // host stubs of the GEMM kernels of the last chunk ([1] = GEMM1, [2] = GEMM2)
};

static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear sticky error for unregistered host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// one device buffer for host-side sources (lkm_create hands the weights over through it chunk by chunk); everything that
// uses it is enqueued on the null stream, so the chunks are ordered without events
struct Staging {
    void* buf = nullptr;
    size_t cap = 0, limit = (size_t)256 << 20;
    Staging() {
        const char* env = getenv("LKM_STAGE_BYTES");
        if (env && atoll(env) > 0) limit = (size_t)atoll(env);
    }
    int reserve(size_t bytes) {
        if (bytes <= cap) return LKM_OK;
        if (buf) {
            (void)hipStreamSynchronize(nullptr);
            (void)hipFree(buf);
            buf = nullptr;
            cap = 0;
        }
        hipError_t e = hipMalloc(&buf, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            buf = nullptr;
            set_error("hipMalloc(%zu) for weight staging failed: %s", bytes, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? LKM_E_NOMEM : LKM_E_HIP;
        }
        cap = bytes;
        return LKM_OK;
    }
    int finish() {
        hipError_t e = hipStreamSynchronize(nullptr);
        if (buf) (void)hipFree(buf);
        buf = nullptr;
        cap = 0;
        if (e != hipSuccess) {
            set_error("weight hand-off failed: %s", hipGetErrorString(e));
            return LKM_E_HIP;
        }
        return LKM_OK;
    }
    ~Staging() {
        if (buf) (void)finish();
    }
};

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

static size_t wbytes_per_elem_x2(int wf) {  // bytes per 2 elements
    switch (wf) {
    case LKM_W_BF16:
    case LKM_W_F16: return 4;
    case LKM_W_FP8_E4M3: return 2;
    default: return 1;
    }
}

// row groups per token tile of the fp8 x fp8 prefill kernel (gemm_prefill_a8w.h: an item takes 8 gate + 8 up tiles of a
// gated GEMM1, 16 tiles otherwise)
static int a8w_row_groups(const LkmEngine* h, int gemm) {
    return gemm == 1 ? ceil_div(h->T1_half, h->gated ? 8 : 16) : ceil_div(h->T2, 16);
}

extern "C" int lkm_abi_version(void) { return LKM_ABI_VERSION; }
extern "C" const char* lkm_last_error(void) { return g_err; }

extern "C" int lkm_device_info(int32_t* n_devices, char* arch_buf, int32_t buf_len) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    if (n_devices) *n_devices = n;
    if (arch_buf && buf_len > 0) {
        arch_buf[0] = 0;
        if (n > 0) {
            hipDeviceProp_t prop;
            LKM_HIP_CHECK(hipGetDeviceProperties(&prop, 0));
            snprintf(arch_buf, buf_len, "%s", prop.gcnArchName);
        }
    }
    return LKM_OK;
}

extern "C" void lkm_destroy(LkmHandle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->w13) (void)hipFree(h->w13);
    if (h->w2) (void)hipFree(h->w2);
    if (h->s13) (void)hipFree(h->s13);
    if (h->s2) (void)hipFree(h->s2);
    if (h->gs13) (void)hipFree(h->gs13);
    if (h->gs2) (void)hipFree(h->gs2);
    if (h->io_x) (void)hipFree(h->io_x);
    if (h->io_ids) (void)hipFree(h->io_ids);
    if (h->io_w) (void)hipFree(h->io_w);
    if (h->io_out) (void)hipFree(h->io_out);
    if (h->pad_x) (void)hipFree(h->pad_x);
    if (h->pad_o) (void)hipFree(h->pad_o);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : h->tune_ev)
        if (e) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int lkm_create(const LkmConfig* cfg, const void* w13, const void* w2,
                          const void* w13_scale, const void* w2_scale, const void* w13_gs,
                          const void* w2_gs, LkmHandle* out) {
    LKM_REQUIRE(cfg && out, "lkm_create: null argument");
    *out = nullptr;
    LKM_REQUIRE(cfg->abi_version == LKM_ABI_VERSION, "lkm_create: ABI version %d != %d", cfg->abi_version, LKM_ABI_VERSION);
    const int wf = cfg->weight_format, adt = cfg->act_dtype;
    LKM_REQUIRE(wf >= LKM_W_BF16 && wf <= LKM_W_MXFP4, "lkm_create: bad weight_format %d", wf);
    LKM_REQUIRE(adt == LKM_DT_BF16 || adt == LKM_DT_F16, "lkm_create: act_dtype must be bf16 or fp16");
    LKM_REQUIRE(!(wf == LKM_W_BF16 && adt != LKM_DT_BF16) && !(wf == LKM_W_F16 && adt != LKM_DT_F16),
                "lkm_create: unquantised weights must have the activation dtype");
    LKM_REQUIRE(w13 && w2, "lkm_create: null weight pointer");
    LKM_REQUIRE(cfg->expert_num > 0 && cfg->expert_num <= 512, "lkm_create: expert_num=%d out of range (1..512)", cfg->expert_num);
    // hidden_size % 8 != 0 (k = 511 of the reference's grid, tests/kernels/moe/test_moe.py:195-208): unquantised weights
    // only -- the image pads K with zeros to the next 64-k unit anyway, the activations pass through aligned scratch rows
    LKM_REQUIRE(cfg->hidden_size > 0 && (cfg->hidden_size % 8 == 0 || wf == LKM_W_BF16 || wf == LKM_W_F16),
                "lkm_create: hidden_size=%d must be a positive multiple of 8 for quantised weight formats", cfg->hidden_size);
    LKM_REQUIRE(cfg->intermediate_size > 0 && cfg->intermediate_size % 8 == 0, "lkm_create: intermediate_size=%d must be a positive multiple of 8", cfg->intermediate_size);
    LKM_REQUIRE(cfg->top_k > 0, "lkm_create: top_k must be > 0");
    LKM_REQUIRE(cfg->activation_type >= LKM_ACT_SILU && cfg->activation_type <= LKM_ACT_RELU2, "lkm_create: bad activation_type %d", cfg->activation_type);
    const bool gated = cfg->has_gate_proj != 0;
    LKM_REQUIRE(!(gated && cfg->activation_type == LKM_ACT_RELU2), "lkm_create: relu2 experts are non-gated (has_gate_proj must be 0)");
    LKM_REQUIRE(!(!gated && cfg->activation_type != LKM_ACT_RELU2), "lkm_create: non-gated experts support activation_type 2 (relu2) only");
    // groupK >= K of a GEMM means ONE scale group per weight row of that GEMM: the reference hands over
    // max(groupK of w13, groupK of w2) (routed_experts.py:1440-1453), so per-channel scales -- _process_fp8(False) for
    // CompressedTensorsW8A8Fp8MoEMethod (:1381-1383, scales [E, N, 1]) and _process_wna16("channel") -- arrive as
    // groupN = 1, groupK = max(hidden, intermediate).  The group of each GEMM is therefore min(groupK, its K).
    const int gk13 = cfg->groupK > cfg->hidden_size ? cfg->hidden_size : cfg->groupK;             // GEMM1: K = hidden
    const int gk2 = cfg->groupK > cfg->intermediate_size ? cfg->intermediate_size : cfg->groupK;   // GEMM2: K = intermediate
    if (wf == LKM_W_FP8_E4M3) {
        LKM_REQUIRE(cfg->fp8_mode == LKM_FP8_W8A16 || cfg->fp8_mode == LKM_FP8_W8A8, "lkm_create: bad fp8_mode %d", cfg->fp8_mode);
        LKM_REQUIRE(cfg->fp8_mode != LKM_FP8_W8A8 || cfg->groupK == 128, "lkm_create: fp8 W8A8 quantises activations in 1x128 groups; groupK must be 128 (got %d)", cfg->groupK);
        LKM_REQUIRE(w13_scale && w2_scale, "lkm_create: fp8 weights need scales");
        LKM_REQUIRE(cfg->groupN > 0 && cfg->groupK > 0, "lkm_create: fp8 needs groupN>0 and groupK>0 (got %d,%d)", cfg->groupN, cfg->groupK);
        LKM_REQUIRE((gk13 % 128 == 0 || gk13 == cfg->hidden_size) && (gk2 % 128 == 0 || gk2 == cfg->intermediate_size),
                    "lkm_create: fp8 groupK=%d must be a multiple of 128 or cover the whole K of a GEMM (hidden %d, intermediate %d)",
                    cfg->groupK, cfg->hidden_size, cfg->intermediate_size);
    }
    if (wf == LKM_W_MXFP4 || wf == LKM_W_NVFP4) {
        const int g = wf == LKM_W_MXFP4 ? 32 : 16;
        LKM_REQUIRE(w13_scale && w2_scale, "lkm_create: fp4 weights need block scales");
        LKM_REQUIRE(cfg->groupK == g && cfg->groupN == 1, "lkm_create: %s block scales are 1 x %d (got groupN=%d groupK=%d)", wf == LKM_W_MXFP4 ? "MXFP4" : "NVFP4", g, cfg->groupN, cfg->groupK);
        LKM_REQUIRE(cfg->hidden_size % 32 == 0 && cfg->intermediate_size % 32 == 0, "lkm_create: fp4 formats need hidden and intermediate sizes that are multiples of 32");
    }
    if (wf == LKM_W_INT4_B8) {
        LKM_REQUIRE(w13_scale && w2_scale, "lkm_create: int4 weights need scales");
        LKM_REQUIRE(cfg->groupN == 1, "lkm_create: int4 expects groupN == 1 (got %d)", cfg->groupN);
        for (int which = 0; which < 2; ++which) {
            const int g = which ? gk2 : gk13, kk = which ? cfg->intermediate_size : cfg->hidden_size;
            LKM_REQUIRE(g >= 32 && (g <= 128 ? 128 % g == 0 : g % 128 == 0), "lkm_create: int4 groupK=%d (group of GEMM%d: %d) unsupported (32, 64, 128 or a multiple of 128)", cfg->groupK, which + 1, g);
            LKM_REQUIRE(kk % g == 0, "lkm_create: int4 group %d must divide K=%d of GEMM%d", g, kk, which + 1);
        }
        LKM_REQUIRE((gk13 >= 128) == (gk2 >= 128) && (gk13 >= 128 || gk13 == gk2), "lkm_create: int4 groups of the two GEMMs (%d, %d) need the same scales per 128-k unit", gk13, gk2);
        const int g = gk13 < gk2 ? gk13 : gk2;
        LKM_REQUIRE(cfg->int4_mode == LKM_INT4_EXACT || cfg->int4_mode == LKM_INT4_FAST || cfg->int4_mode == LKM_INT4_ZP, "lkm_create: bad int4_mode %d", cfg->int4_mode);
        LKM_REQUIRE(cfg->int4_mode != LKM_INT4_ZP || (w13_gs && w2_gs), "lkm_create: int4_mode ZP takes the zero points (uint8 [E, rows, K / group]) in the two global-scale pointer slots");
        LKM_REQUIRE(cfg->int4_mode != LKM_INT4_FAST || g % 128 == 0, "lkm_create: int4_mode FAST applies the group scale per 128-k block; groupK must be a multiple of 128 (got %d)", g);
    }
    int ndev = 0;
    LKM_HIP_CHECK(hipGetDeviceCount(&ndev));
    LKM_REQUIRE(cfg->gpu_id >= 0 && cfg->gpu_id < ndev, "lkm_create: gpu_id=%d but %d HIP devices visible", cfg->gpu_id, ndev);
    LKM_HIP_CHECK(hipSetDevice(cfg->gpu_id));
    {
        hipDeviceProp_t prop;
        LKM_HIP_CHECK(hipGetDeviceProperties(&prop, cfg->gpu_id));
        LKM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "lkm_create: device %d is %s; this engine is built for gfx950 (MI355X) only", cfg->gpu_id, prop.gcnArchName);
    }

    LkmEngine* h = new LkmEngine();
    h->cfg = *cfg;
    h->device = cfg->gpu_id;
    h->E = cfg->expert_num;
    h->Hu = cfg->hidden_size;
    h->H = round_up(cfg->hidden_size, 8);
    h->I = cfg->intermediate_size;
    h->K = cfg->top_k;
    h->gated = gated;
    h->interleaved = gated && cfg->activation_type == LKM_ACT_SWIGLUOAI;
    h->wf = wf;
    h->adt = adt;
    h->a8 = wf == LKM_W_FP8_E4M3 && cfg->fp8_mode == LKM_FP8_W8A8;
    h->ps = wf == LKM_W_INT4_B8 && cfg->int4_mode == LKM_INT4_FAST;
    h->zp = wf == LKM_W_INT4_B8 && cfg->int4_mode == LKM_INT4_ZP;
    h->wfk = h->a8 ? LKM_W_FP8_A8 : (h->ps ? LKM_W_INT4_PS : (h->zp ? LKM_W_INT4_ZP : wf));
    h->unitk = wf_unitk(wf);
    h->T1_half = round_up(ceil_div(h->I, 16), 4);
    h->U1 = ceil_div(h->H, h->unitk);
    h->T2 = round_up(ceil_div(h->H, 16), 4);
    h->U2 = ceil_div(h->I, h->unitk);
    h->ld_act = h->I;
    h->spu = (wf == LKM_W_INT4_B8) ? (gk13 >= 128 ? 1 : 128 / gk13) : 0;

    const int loads = wf_loads(wf);
    const int halves = gated ? 2 : 1;
    const size_t w13_vec = (size_t)h->E * halves * h->T1_half * h->U1 * loads * 64;
    const size_t w2_vec = (size_t)h->E * h->T2 * h->U2 * loads * 64;
    int rc = LKM_OK;
    auto fail = [&](int code) {
        lkm_destroy(h);
        (void)hipGetLastError();     // a failed hipMalloc stays the "last error": the next launch check must not see it
        return code;
    };
#define LKM_TRY(expr)                        \
    do {                                     \
        rc = (expr);                         \
        if (rc != LKM_OK) return fail(rc);   \
    } while (0)
#define LKM_TRY_HIP(expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                       \
            return fail(_e == hipErrorOutOfMemory ? LKM_E_NOMEM : LKM_E_HIP);               \
        }                                                                                   \
    } while (0)

    LKM_TRY_HIP(hipMalloc(&h->w13, w13_vec * 16));
    LKM_TRY_HIP(hipMalloc(&h->w2, w2_vec * 16));
    h->weight_bytes = (int64_t)(w13_vec + w2_vec) * 16;

    h->loads = loads;
    // (source dims are the caller's: K = Hu for w13, Hu rows for w2 -- the repack kernels bound every element read)
    RepackDims d13{h->E, h->I, halves, h->interleaved ? 1 : 0, h->Hu, h->T1_half, h->U1, h->a8 ? 1 : 0};
    RepackDims d2{h->E, h->Hu, 1, 0, h->I, h->T2, h->U2, h->a8 ? 1 : 0};
    // hand-off of the caller's tensors: device sources are read in place; host sources pass through ONE
    // staging buffer of <= LKM_STAGE_BYTES (default 256 MiB, never less than one expert) in chunks of whole experts --
    // copy, repack, next chunk, all in stream order -- so creation adds at most one chunk to the footprint of the
    // finished image, and the host waits once, at the end (`staging` synchronises and frees when it goes out of scope).
    Staging staging;
    // src: [E][src_pe bytes]; dst: [E][dst_pe bytes]; launch(src chunk, dst chunk, dims with E = experts in the chunk)
    auto hand_off = [&](const void* src, size_t src_pe, void* dst, size_t dst_pe, RepackDims d, auto&& launch) -> int {
        if (is_device_ptr(src)) return launch(src, dst, d);
        size_t per = staging.limit / (src_pe ? src_pe : 1);
        per = per < 1 ? 1 : (per > (size_t)h->E ? (size_t)h->E : per);
        int r = staging.reserve(per * src_pe);
        for (size_t e0 = 0; r == LKM_OK && e0 < (size_t)h->E; e0 += per) {
            const size_t ne = (size_t)h->E - e0 < per ? (size_t)h->E - e0 : per;
            hipError_t ce = hipMemcpyAsync(staging.buf, (const char*)src + e0 * src_pe, ne * src_pe, hipMemcpyHostToDevice, nullptr);
            if (ce != hipSuccess) {
                set_error("hipMemcpy H2D of %zu weight bytes failed: %s", ne * src_pe, hipGetErrorString(ce));
                return LKM_E_HIP;
            }
            d.E = (int)ne;
            r = launch(staging.buf, (char*)dst + e0 * dst_pe, d);
        }
        return r;
    };
    const int wfr = h->ps ? LKM_W_INT4_PS : wf;
    {
        const size_t b13 = (size_t)halves * h->I * h->Hu * wbytes_per_elem_x2(wf) / 2;    // source bytes per expert
        const size_t b2 = (size_t)h->Hu * h->I * wbytes_per_elem_x2(wf) / 2;
        auto rw = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_w(nullptr, wfr, sp, dp, dd); };
        LKM_TRY(hand_off(w13, b13, h->w13, w13_vec / h->E * 16, d13, rw));
        LKM_TRY(hand_off(w2, b2, h->w2, w2_vec / h->E * 16, d2, rw));
    }
    if (h->ps) {     // fp32 scale per (row, 128-k unit), the layout of the fp8 block scales
        const size_t n13 = (size_t)halves * h->T1_half * h->U1 * 16, n2 = (size_t)h->T2 * h->U2 * 16;   // per expert
        LKM_TRY_HIP(hipMalloc(&h->s13, h->E * n13 * 4));
        LKM_TRY_HIP(hipMalloc(&h->s2, h->E * n2 * 4));
        h->weight_bytes += (int64_t)h->E * (n13 + n2) * 4;
        auto rs13 = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_int4ps(nullptr, sp, dp, dd, gk13, adt); };
        auto rs2 = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_int4ps(nullptr, sp, dp, dd, gk2, adt); };
        LKM_TRY(hand_off(w13_scale, (size_t)halves * h->I * (h->H / gk13) * 2, h->s13, n13 * 4, d13, rs13));
        LKM_TRY(hand_off(w2_scale, (size_t)h->H * (h->I / gk2) * 2, h->s2, n2 * 4, d2, rs2));
    } else if (h->zp) {     // (scale, zero point) pairs: two passes over one image, the zero points from the global-scale slots
        const size_t n13 = (size_t)halves * h->T1_half * h->U1 * 16 * h->spu, n2 = (size_t)h->T2 * h->U2 * 16 * h->spu;
        LKM_TRY_HIP(hipMalloc(&h->s13, h->E * n13 * 4 + 16));   // +16: the kernels fetch 16 bytes per lane
        LKM_TRY_HIP(hipMalloc(&h->s2, h->E * n2 * 4 + 16));
        h->weight_bytes += (int64_t)h->E * (n13 + n2) * 4;
        for (int which = 0; which < 2; ++which) {
            auto rs13 = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_int4zp(nullptr, sp, dp, dd, gk13, h->spu, which, adt); };
            auto rs2 = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_int4zp(nullptr, sp, dp, dd, gk2, h->spu, which, adt); };
            const size_t eb = which ? 1 : 2;       // source bytes per entry: act-dtype scale / uint8 zero point
            LKM_TRY(hand_off(which ? w13_gs : w13_scale, (size_t)halves * h->I * (h->H / gk13) * eb, h->s13, n13 * 4, d13, rs13));
            LKM_TRY(hand_off(which ? w2_gs : w2_scale, (size_t)h->H * (h->I / gk2) * eb, h->s2, n2 * 4, d2, rs2));
        }
    } else if (wf == LKM_W_INT4_B8) {
        const size_t n13 = (size_t)halves * h->T1_half * h->U1 * 16 * h->spu, n2 = (size_t)h->T2 * h->U2 * 16 * h->spu;
        LKM_TRY_HIP(hipMalloc(&h->s13, h->E * n13 * 2 + 16));   // +16: the kernels fetch 8 bytes per lane
        LKM_TRY_HIP(hipMalloc(&h->s2, h->E * n2 * 2 + 16));
        h->weight_bytes += (int64_t)h->E * (n13 + n2) * 2;
        auto rs13 = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_int4(nullptr, sp, dp, dd, gk13, h->spu); };
        auto rs2 = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_int4(nullptr, sp, dp, dd, gk2, h->spu); };
        LKM_TRY(hand_off(w13_scale, (size_t)halves * h->I * (h->H / gk13) * 2, h->s13, n13 * 2, d13, rs13));
        LKM_TRY(hand_off(w2_scale, (size_t)h->H * (h->I / gk2) * 2, h->s2, n2 * 2, d2, rs2));
    } else if (wf == LKM_W_MXFP4 || wf == LKM_W_NVFP4) {
        const int g = cfg->groupK, spu = 128 / g;
        const size_t n13 = (size_t)halves * h->T1_half * h->U1 * 16 * spu, n2 = (size_t)h->T2 * h->U2 * 16 * spu;
        LKM_TRY_HIP(hipMalloc(&h->s13, h->E * n13));
        LKM_TRY_HIP(hipMalloc(&h->s2, h->E * n2));
        h->weight_bytes += (int64_t)h->E * (n13 + n2);
        const int pad = wf == LKM_W_MXFP4 ? 127 : 0x38;   // 1.0 in E8M0 / e4m3fn (the padded weights are 0)
        auto rs = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_fp4(nullptr, sp, dp, dd, g, pad); };
        LKM_TRY(hand_off(w13_scale, (size_t)halves * h->I * (h->H / g), h->s13, n13, d13, rs));
        LKM_TRY(hand_off(w2_scale, (size_t)h->H * (h->I / g), h->s2, n2, d2, rs));
        if (wf == LKM_W_NVFP4) {   // per-expert multipliers (NULL = 1.0)
            for (int which = 0; which < 2; ++which) {
                const void* src = which ? w2_gs : w13_gs;
                float** dst = which ? &h->gs2 : &h->gs13;
                if (!src) continue;
                LKM_TRY_HIP(hipMalloc(dst, (size_t)h->E * 4));
                LKM_TRY_HIP(hipMemcpyAsync(*dst, src, (size_t)h->E * 4, hipMemcpyDefault, nullptr));
            }
        }
    } else if (wf == LKM_W_FP8_E4M3) {
        const int gN = cfg->groupN, gK = cfg->groupK;
        const size_t n13 = (size_t)halves * h->T1_half * h->U1 * 16, n2 = (size_t)h->T2 * h->U2 * 16;
        LKM_TRY_HIP(hipMalloc(&h->s13, h->E * n13 * 4));
        LKM_TRY_HIP(hipMalloc(&h->s2, h->E * n2 * 4));
        h->weight_bytes += (int64_t)h->E * (n13 + n2) * 4;
        const size_t src13 = (size_t)ceil_div(halves * h->I, gN) * ceil_div(h->H, gK) * 4;
        const size_t src2 = (size_t)ceil_div(h->H, gN) * ceil_div(h->I, gK) * 4;
        auto rs = [&](const void* sp, void* dp, const RepackDims& dd) { return launch_repack_s_fp8(nullptr, sp, dp, dd, gN, gK); };
        LKM_TRY(hand_off(w13_scale, src13, h->s13, n13 * 4, d13, rs));
        LKM_TRY(hand_off(w2_scale, src2, h->s2, n2 * 4, d2, rs));
    }
    LKM_TRY(staging.finish());     // the one host wait of the hand-off; the caller may free its tensors after this

    // scratch: sized for the larger of the decode batch and one prefill chunk
    size_t chunk = cfg->group_max_len > 0 ? (size_t)cfg->group_max_len : 4224;
    if (cfg->max_batch_size > 0 && (size_t)cfg->max_batch_size < chunk) chunk = cfg->max_batch_size;
    size_t cap = cfg->max_num_seqs > 0 ? (size_t)cfg->max_num_seqs : 1;
    if (chunk > cap) cap = chunk;
    h->cap_tokens = cap;
    const size_t slots = cap * h->K;
    const size_t y_rows = slots > 8 * (slots < 2048 ? slots : 2048) ? slots : 8 * (slots < 2048 ? slots : 2048);
    {
        const size_t kb1 = ceil_div(h->H, 128), kb2 = ceil_div(h->I, 128);
        const bool xs = h->a8 || h->ps;       // per-(row, 128-k) scalars: activation scales / activation sums
        // the fp8 x fp8 prefill kernel's item lists: (256-row token tiles of the largest chunk) x (row groups of GEMM1 + GEMM2)
        const size_t items_n = h->a8 ? (slots / 256 + (size_t)h->E + 8) * (size_t)(a8w_row_groups(h, 1) + a8w_row_groups(h, 2)) * 4 : 0;
        LKM_TRY(arena_reserve(h->device, h->E, slots, slots * h->ld_act, y_rows * h->H,
                              h->a8 ? cap * h->H : 0, xs ? cap * kb1 : 0, h->a8 ? slots * h->I : 0,
                              xs ? slots * kb2 : 0, items_n, &h->arena));
    }
    if (h->Hu != h->H) {       // aligned scratch rows for the token and output matrices of one chunk
        LKM_TRY_HIP(hipMalloc(&h->pad_x, cap * (size_t)h->H * 2));
        LKM_TRY_HIP(hipMalloc(&h->pad_o, cap * (size_t)h->H * 4));
        h->pad_tokens = cap;
    }
    for (auto& e : h->ev) LKM_TRY_HIP(hipEventCreate(&e));
    *out = h;
    return LKM_OK;
#undef LKM_TRY
#undef LKM_TRY_HIP
}

// ------------------------------------------------------------------ launch geometry heuristics
// Launch plan of one step.  skinny: weight streamer, token operand straight from L2 (few rows per
// expert).  tiled: token operand staged through LDS (many rows per expert).  hybrid: both kernels run,
// the skinny one skips experts with more than `split_rows` rows and the tile list holds only those.
struct Plan {
    LaunchCfg s1, s2;   // skinny GEMM1 / GEMM2 (s1.tb == 0: not launched)
    LaunchCfg t1, t2;   // tiled  GEMM1 / GEMM2 (t1.tiled == 0: not launched)
    int split_rows;     // hybrid threshold (0: no split)
    int xcd1, xcd2;     // GEMM1 / GEMM2: XCD-aware work mapping (gemm_tiled.h, dispatch.hip: per-XCD tile runs)
    // mixed tile heights (round 5): experts with more than big_min rows are cut into big_rows-row tiles and run on b1 / b2,
    // the others keep t1 / t2 -- two launches per GEMM on the two parts of one tile list (big_rows == 0: off)
    int big_rows, big_min;
    LaunchCfg b1, b2;
};

static void pick_cfg(const LkmEngine* h, int M, size_t n_slots, Plan* pl) {   // M, n_slots: as handed over (incl. -1 slots)
    // Measured on MI355X (profiles/r01_sweep_*): one 16-row tile per wave streams best (7168 waves x 64
    // threads for Mixtral GEMM1: 6.7 TB/s vs 6.4 TB/s at two tiles); two tiles only for sub-16-bit
    // weights, where the token operand dominates the load instructions.
    const int kMinWaves = 2048;
    // Expert-parallel callers hand over slot lists in which only ~1/ep of the ids are local (the rest are
    // -1: fixed-capacity all-to-all, or the reference's replicated-token mode); the sort drops them on
    // the device, but the host-side plan must not size tiles for rows that will not exist.  `valid_den`
    // (lkm_set_tuning) = that ep; 0/1 = every slot counts.
    const size_t n_eff = h->t_valid_den > 1 ? (n_slots + h->t_valid_den - 1) / h->t_valid_den : n_slots;
    const int n_act = (int)((size_t)h->E < n_eff ? (size_t)h->E : n_eff);
    const size_t avg_rows = n_eff / (size_t)(n_act > 0 ? n_act : 1);
    if (h->t_valid_den > 1) M = (int)((size_t)M < n_eff ? (size_t)M : n_eff);
    // token blocks held in registers: sized by the rows an expert is LIKELY to get (2x the mean + 8),
    // not by M; a rare fuller expert goes to the tiled kernel (hybrid) or loops super-blocks.
    // (DSv3 slice, 256 rows over 32 experts: tb=4 376 us, tb=2 316 us, tb=1 306 us.)
    const size_t est_max = (size_t)M < 2 * avg_rows + 8 ? (size_t)M : 2 * avg_rows + 8;
    int tb = est_max <= 16 ? 1 : (est_max <= 32 ? 2 : 4);
    if (h->t_tb > 0) tb = h->t_tb;

    // ---- which kernels
    // Mixtral bf16: M=64 (16 rows/expert) skinny 537 us vs tiled 570 us; M=128 (32 rows/expert) skinny
    // 714 us vs tiled64 585 us; 64-row tiles beat 128-row tiles up to M=512, 4 waves beat 8.
    const bool w16 = h->wf == LKM_W_BF16 || h->wf == LKM_W_F16;
    // gemm_prefill.h ("pf" = 8, 256-row tiles; 16-bit weights and fp8 W8A16): K loops of whole 128-k blocks, tokens, the
    // intermediate and an expert's weights inside 2 GiB buffer windows (prefill_kernel_ok); fp8: one scale per 16-row
    // tile and 128-k unit
    const bool w8a16 = h->wf == LKM_W_FP8_E4M3 && !h->a8;
    // ... and the 4-bit formats (decoded once per workgroup into the 16-bit image): uint4b8, MXFP4, NVFP4
    const bool w4pf = (h->wf == LKM_W_INT4_B8 && !h->ps && (!h->zp || h->spu <= 2)) || h->wf == LKM_W_MXFP4 || h->wf == LKM_W_NVFP4;   // (not the fast int4 mode: its own image)
    const size_t pf_ub = w4pf ? 1024 : 2048;     // bytes of a (tile, unit) of the image
    const bool pf8_ok = (w16 || w8a16 || w4pf) && h->H % 128 == 0 && h->I % 128 == 0 && (size_t)M * (size_t)h->H * 2 < (size_t)0x7fffffff &&
                        n_slots * (size_t)h->ld_act * 2 < (size_t)0x7fffffff &&
                        (size_t)h->T1_half * (h->gated ? 2 : 1) * h->U1 * pf_ub < (size_t)0x7fffffff &&
                        (size_t)h->T2 * h->U2 * pf_ub < (size_t)0x7fffffff &&
                        (!w4pf || ((size_t)h->E * h->T1_half * (h->gated ? 2 : 1) * h->U1 * 128 < (size_t)0x7fffffff &&      // (scale bytes: 32-bit offsets)
                                   (size_t)h->E * h->T2 * h->U2 * 128 < (size_t)0x7fffffff)) &&
                        (!w8a16 || ((size_t)h->E * h->T1_half * (h->gated ? 2 : 1) * h->U1 * 16 < (size_t)0x7fffffff &&
                                    (size_t)h->E * h->T2 * h->U2 * 16 < (size_t)0x7fffffff && h->U1 <= 64 && h->U2 <= 64 &&
                                    // block scales shared by whole 16-row tiles (carried in the accumulators), or per-row scales
                                    // that do not change along K (per-channel: applied to the finished accumulators)
                                    ((h->cfg.groupN % 16 == 0 && h->cfg.groupK % 128 == 0) ||
                                     (h->cfg.groupK >= h->H && h->cfg.groupK >= h->I))));
    // (256-row tiles of fp8-W8A16 / 4-bit weights exist on gemm_prefill.h only -- "pf" = 0 or 8; with any other
    //  "pf" forced, e.g. 5 / 6 = the decode kernels of gemm_w4x.h, prefill-sized chunks keep the 64 / 128-row tiles)
    const bool pf8_sel = pf8_ok && (h->t_pf == 0 || h->t_pf == 8);
    int tiled = 0, split = 0, g2_only = 0;
    {
        // Tile rows by rows per expert (profiles/r01_tile_thresholds.log, Mixtral shapes, 8 experts):
        // 16-bit weights 48 rows/expert: 64 (507 us vs 604 at 128); 64: 128 (602 vs 746); 96: 128 (628 vs
        // 655 at 256); 128: 256 (735 vs 862); 256: 256 (973 vs 1149).  MXFP4: 128 from 64 rows/expert
        // (325 vs 354 us).  fp8-W8A16 / int4 stay at 64 (the 128-row variants of the formats that decode
        // in registers run out of them); fp8-W8A8: see below.
        if (M > 32 && avg_rows >= 16) {   // bf16 M=64: 497 (streamer) vs 466 us, M=96: 562 (hybrid) vs 467; M=48 stays hybrid
            tiled = 64;
            if (w16 && avg_rows > 56) tiled = avg_rows >= 112 ? 256 : 128;
            // many experts (GLM-4.5-Air, 128): the 256-row kernel from 64 rows per expert (its narrow loop takes the tiles of
            // <= 128 tokens) -- bf16 M=1024 / 1280 / 1536: 886 -> 872, 924 -> 896, 950 -> 915 us; fp8-W8A16 742 -> 689 (M=1024),
            // 983 -> 752 (M=1536); Mixtral's 8 experts keep 128-row tiles below 112 rows (M=384: 572 vs 609 us):
            // profiles/r04_prefill16_threshold.log
            if (((w16 && h->t_pf >= 0 && pf8_ok) || (w8a16 && pf8_sel)) && avg_rows >= 64 && n_act >= 32) tiled = 256;
            // fp8 W8A16 at prefill sizes (MOE_FP8.gpu_prefill): 256-row tiles on gemm_prefill.h (round 4: raw fp8 through the
            // LDS-DMA ring, converted in registers, block scales carried in the accumulators) where 16-bit weights take them;
            // needs one scale per 16-row tile and 128-k unit (block heights in multiples of 16: pf8_ok)
            if (w8a16 && avg_rows >= 112 && pf8_sel) tiled = 256;
            if (h->wf == LKM_W_MXFP4 && avg_rows >= 64) tiled = 128;
            // uint4b8 / NVFP4 at prefill sizes (round 4, profiles/r04_prefill_plan_sweep.log: what MOE_WNA16 / MOE_NVFP4
            // gpu_prefill runs): 128-row tiles x 8 waves from ~100 rows per expert -- int4 Mixtral M=512 725 -> 683 us,
            // M=2048 2159 -> 1951, M=8192 8235 -> 7149 (GEMM1 -21 %); NVFP4 M=2048 2017 -> 1842.  Decode sizes keep 64 / 32.
            if ((h->wf == LKM_W_INT4_B8 || h->wf == LKM_W_NVFP4) && avg_rows >= 96) tiled = 128;
            // ... and from 112 rows per expert the 256-row prefill kernel with the weights decoded once per workgroup
            // (gemm_prefill.h W4, round 4; Mixtral M=1024 / 2048 / 4096 / 8192, step us: uint4b8 1100 -> 926, 1953 -> 1726, 3696 ->
            // 3068, 7170 -> 5758; NVFP4 M=8192 6677 -> 5723; MXFP4 M=8192 5340 -> 4583: profiles/r04_prefill16_w4.log)
            // (many experts: from 192 rows -- GLM-4.5-Air int4, 128 experts: M=2048 / 128 rows 1139 -> 1189 us, M=8192 3687 -> 2864;
            // the 4-bit mode has no narrow loop for the partial tiles a wide row-count distribution leaves)
            if (w4pf && avg_rows >= (n_act >= 32 ? 192u : 112u) && pf8_sel) tiled = 256;
            // fp8 x fp8 (W8A8): 128-row tiles from ~200 rows per expert, now that the per-unit partial sums
            // are formed four token blocks at a time and the kernel fits its registers (GLM-4.5-Air prefill
            // 3677 -> 3179 us with two GEMM2 tiles per wave; Mixtral M=2048 1996 -> 1820; M=512 equal)
            if (h->a8 && avg_rows >= 192) tiled = 128;
            // ... and from there on the persistent prefill kernel (gemm_prefill_a8w.h: 256 x 256 items, one plain
            // v_mfma_f32_16x16x128_f8f6f4 per 128-k block) where the shape qualifies
            // (the kernels address tokens, token scales and an expert's weights through 2 GiB buffer windows; a chunk
            // that does not fit them keeps the 128-row tiles instead of failing the step: prefill_a8w_ok)
            const bool win_ok = n_slots < ((size_t)1 << 22) &&            // (prefill_a8w_ok: GEMM1's slot -> token division)
                                (size_t)M * (size_t)h->H < (size_t)0x7fffffff &&
                                n_slots * (size_t)h->ld_act < (size_t)0x7fffffff &&
                                (size_t)h->T1_half * (h->gated ? 2 : 1) * h->U1 * 2048 < (size_t)0x7fffffff &&
                                (size_t)h->T2 * h->U2 * 2048 < (size_t)0x7fffffff;
            if (h->a8 && avg_rows >= 192 && h->t_pf >= 0 && h->H % 128 == 0 && h->I % 128 == 0 &&
                h->U1 >= 8 && h->U2 >= 8 && h->U1 <= 64 && h->U2 <= 64 &&
                h->cfg.groupN % 16 == 0 && h->cfg.groupK == 128 && win_ok)
                tiled = 256;
        } else if (h->wf == LKM_W_FP8_E4M3 && M >= 48) {
            // fp8 (both modes): tiles from 48 tokens on (profiles/r01_fp8_tile_threshold.log: Mixtral W8A8
            // M=48 275 -> 265 us, M=64 356 -> 273, M=96 379 -> 281; DSv3 rank slice, 256 rows over 32
            // experts, 280 -> 259; M=16/32 stay with the streamer, 248 vs 261 and 254 vs 260)
            tiled = 64;
        } else if (wf_is_4bit(h->wf) && M >= 8) {
            // 4-bit weights: the tile kernel (token rows staged once per workgroup through LDS, 4 waves x
            // 3 resident workgroups) beats the streamer at EVERY decode batch measured, Mixtral shapes
            // (profiles/r01_4bit_tile_threshold.log): MXFP4 M=8 148 -> 124 us, M=32 166 -> 155, M=64 267 ->
            // 170; int4 M=16 227 -> 196, M=64 300 -> 223; NVFP4 M=32 200 -> 186, M=64 296 -> 213.
            tiled = 64;
        } else if (M > 16 * tb && h->t_hybrid >= 0) {
            tiled = 64;
            split = 16 * tb;
            // 16-bit weights (round 5, profiles/r05_hybrid_vs_tiles_ab.log): plain 64-row tiles instead of the hybrid -- under
            // Zipf routing the hybrid's streamer half re-reads the hot experts' token rows per weight tile and the step loses
            // 6-13 % (DSv3 bf16 rank slice M=64 / 128 / 256: 211 -> 188, 290 -> 253, 327 -> 288 us; Qwen3-30B-A3B M=32 / 64 /
            // 128: 127 -> 112, 155 -> 140, 184 -> 173; Mixtral M=48: 544 -> 473), under uniform routing the two are within
            // +-3 % (465 -> 455, 442 -> 439, 388 -> 398; 204 -> 201, 228 -> 224, 244 -> 231; 471 -> 470).  "hybrid" = 1 keeps
            // the hybrid.
            if (w16 && h->t_hybrid == 0) split = 0;
            // fp8 (both modes) below 48 tokens (profiles/r05_plan_robustness_sweep.log, r05_plan_sweep_followup.log): 32-row
            // tiles -- Mixtral W8A8 M=40: 261 -> 256 us uniform, 296 -> 263 Zipf; DSv3 rank slice (32 experts) W8A8 M=32:
            // 168 -> 174 uniform, 132 -> 98 Zipf; W8A16 M=32 / 40: 172 -> 167 / 183 -> 173 uniform, 137 -> 93 / 151 -> 115
            // Zipf.  The streamer half of the hybrid is what skew hurts; no default plan uses the hybrid any more.
            if (h->wf == LKM_W_FP8_E4M3 && h->t_hybrid == 0) {
                tiled = 32;
                split = 0;
            }
        }
        // 4-bit decode batches in which an expert is unlikely to hold more than 32 rows: the 32-row tile keeps half
        // the accumulators and two more waves per SIMD resident (profiles/r01_tile32.log: int4 M=16 196 -> 186 us,
        // M=32 196 -> 190, MXFP4 M=32 158 -> 150).  fp8 stays at 64: the DSv3 rank slice loses in GEMM1 (147 ->
        // 160 us) what it gains in GEMM2 (88 -> 85), Mixtral W8A8 M=48 is equal.
        if (tiled == 64 && !split && wf_is_4bit(h->wf) && est_max <= 32) tiled = 32;
        // fp8 x fp8 with many experts (round 3, with the four-deep weight ring below): the 32-row tile for the DSv3 rank
        // slice, uniform 266 -> 265 us per step, Zipf (18 of 32 experts hit, the hot one 126 rows) 196 -> 186 (GEMM1 112 ->
        // 102 us: twice the workgroups on a chip the 64-row grid leaves under-subscribed).  W8A16 keeps 64 rows (uniform 263
        // vs 266, Zipf 189 vs 200): profiles/r03_mixed_plan_sweep.log
        if (tiled == 64 && !split && h->a8 && est_max <= 32 && n_act >= 16 &&
            (long long)n_act * ((h->T1_half + 3) / 4) <= 4096)      // (an under-subscribed grid only: all 256 DSv3 experts on one GPU
            tiled = 32;                                             //  under Zipf routing LOSE 10 %, 840 -> 927 us, uniform equal)
        if (h->t_tiled > 0) { tiled = h->t_tiled; split = 0; }
        if (h->t_tiled < 0) { tiled = 0; split = 0; }
        // Mixed plan (streamer formats, an expert may hold more than one token block): GEMM1 stays with the streamer, GEMM2
        // takes the tile kernel.  A streamer wave re-reads its expert's token rows from the L2 for every weight tile, and
        // GEMM2's rows are I = 3.5 x H long: with skewed routing the expert that holds most of the rows (Mixtral M=32, Zipf:
        // 23 of 64) costs 660 KB of L2 reads per 16-row weight tile, where the tile kernel stages the rows through LDS once
        // per four waves.  Mixtral M=32, step us, streamer GEMM2 -> tile GEMM2: bf16 uniform 455 -> 461, Zipf 477 -> 460;
        // fp8-W8A8 uniform 257 -> 255, Zipf 289 -> 266 (32-row tiles); bf16 M=20 / 24 gain nothing and pay the tile list
        // (+4 us), so the rule starts where two full token blocks are likely (profiles/r03_zipf_launch_order_sweep.log,
        // r03_mixed_plan_sweep.log).  "tiled2" = -1 switches it off, n > 0 forces n-row GEMM2 tiles.
        if (!tiled && !split && (w16 || h->wf == LKM_W_FP8_E4M3) && est_max >= 24 && h->t_tiled2 >= 0 && h->t_tiled >= 0 &&
            h->t_fuse == 0)            // ("fuse" = +-1 asks for the streamer GEMM2 with / without the combine folded in)
            g2_only = 32;          // (the rule only holds while an expert cannot exceed 32 rows: one 32-row tile each; against
                                   //  64-row tiles no measurable difference, profiles/r03_headline_gemm2_ab.log)
        if (!tiled && !split && h->t_tiled2 > 0) g2_only = h->t_tiled2;
        if (g2_only) tiled = g2_only;
    }
    pl->split_rows = split;
    pl->big_rows = pl->big_min = 0;
    pl->b1 = pl->b2 = LaunchCfg{0, 0, 0, 0, 0, 0, 0, 0};
    // XCD-aware work mapping (gemm_tiled.h; the workgroups that share a token tile or a weight panel run on
    // ONE XCD at the same time) stays a knob ("xcd"): bytes that miss the L2 arrive at <= 7.6 TB/s chip-wide
    // against 32 TB/s from the L2 (tools/probe_l2.hip), and the mapping lifts GLM-4.5-Air GEMM1's L2 hits
    // from 19 to 54 % -- but the step gains 2 % with uniform routing and LOSES 7 % with Zipf routing (equal
    // item counts per XCD are not equal work), GEMM2 +6 %, Mixtral M=4096 +27 %: profiles/r01_prefill_pmc.md.
    pl->xcd1 = pl->xcd2 = 0;
    pl->t1 = pl->t2 = LaunchCfg{0, 0, 0, 0, 0, 0, 0, 0};
    pl->s1 = pl->s2 = LaunchCfg{0, 0, 0, 0, 0, 0, 0, 0};
    if (tiled) {
        // 128-row tiles with 16-bit weights: 4-wave workgroups when 8-wave ones would not cover the chip
        // twice (few experts per rank: EP=8 Mixtral has 1 expert = 112 eight-wave workgroups)
        const long long wg8 = (long long)n_act * ((avg_rows + tiled - 1) / tiled) * ((h->T1_half + 7) / 8);
        int waves = tiled >= 128 ? 8 : 4;
        if (tiled == 128 && (h->wf == LKM_W_BF16 || h->wf == LKM_W_F16) && wg8 < 512) waves = 4;
        if (h->t_waves > 0) waves = h->t_waves;
        const int nt1 = (h->t_nt1 > 0 && !split && !(tiled == 256 && h->a8)) ? h->t_nt1 : 1;   // (4-bit DMA kernel: 1 or 2)
        // two weight tiles per wave in GEMM2: 256-row tiles (GLM bf16: 3.61 vs 3.81 ms), fp8-W8A8 128-row tiles
        // (GEMM2 1343 -> 1141 us) and the 4-bit formats at 64-row tiles (half the token-fragment LDS reads per
        // weight byte; Mixtral M=128 GEMM2 int4 95 -> 92 us, NVFP4 85.6 -> 80.4, MXFP4 74.2 -> 67.5:
        // profiles/r01_int4_hoist_nt2.log)
        const bool w4_64 = wf_is_4bit(h->wf) && tiled == 64 && waves == 4 && !split;   // (the 8-wave variant exists with one tile only)
        const int nt2 = (tiled == 256 && h->a8) ? 1 : (h->t_nt2 > 0 && !split) ? h->t_nt2 : ((tiled == 256 || (tiled == 128 && h->a8) || w4_64) ? 2 : 1);
        // weight/token register ring depth (64-row tiles; the larger tiles have no registers to spare).
        // Measured at M=128 (profiles/r01_prefetch_depth.log): bf16 4/4 (473 vs 525 us at 2/2); the
        // formats that decode in registers keep GEMM1 at 2 (occupancy): int4 2/4 260 us vs 4/4 291 us,
        // MXFP4 205 vs 221 us, fp8 293 vs 298 us.
        int pd1 = 2, pd2 = 2;
        if (tiled <= 64) {
            const bool w16 = h->wf == LKM_W_BF16 || h->wf == LKM_W_F16;
            pd1 = w16 ? 4 : 2;
            pd2 = 4;
            // fp8 with a short K loop (DSv3: I = 2048 = 16 units): four resident waves beat the deeper ring,
            // GEMM2 W8A16 91.3 -> 83.9 us, W8A8 90.8 -> 88.4
            if (h->wf == LKM_W_FP8_E4M3 && h->U2 <= 32) pd2 = 2;
            if (h->wf == LKM_W_FP8_E4M3 && tiled == 32) pd1 = pd2 = 4;      // (the 32-row tile has the registers for it)
            // int4 / NVFP4 with two GEMM2 tiles per wave: the decoders are VALU-heavy, the resident wave is worth
            // more than the deeper ring (int4 GEMM2 92 -> 84.5 us, NVFP4 80.4 -> 78.6; MXFP4 LOSES: 67.5 -> 72.7)
            if (w4_64 && h->wf != LKM_W_MXFP4) pd2 = 2;
        }
        if (tiled == 128 && waves == 4) pd1 = pd2 = 4;   // EP=8 Mixtral rank: 154 vs 229 us at 2/2
        int pf = (tiled == 256 && h->t_pf > 0 && !wf_is_4bit(h->wf)) ? h->t_pf : 0;
        // 16-bit weights on 256-row tiles: the 8-phase LDS-DMA kernel (gemm_prefill.h, round 4) is the default -- GLM-4.5-Air
        // bf16 prefill M=8192: GEMM1 1674 -> 1324 us, GEMM2 964 -> 728, step 2919 -> 2227 (profiles/r04_prefill16_*.log);
        // "pf" = -1 keeps gemm_tiled_kernel
        if (tiled == 256 && (w16 || w8a16) && h->t_pf == 0) pf = 8;
        if (tiled == 256 && w4pf && (h->t_pf == 0 || h->t_pf == 8)) pf = 8;
        if (pf == 8 && !pf8_ok) pf = 0;
        // (round 2's LDS-DMA ring kernel for the 4-bit formats, "pf" = 4, was removed in round 4: three to four times the
        // bytes in flight per CU and no faster, profiles/r02_w4dma_sweep.log)
        // round 4: the 32x32-MFMA kernels of the 4-bit formats (gemm_w4x.h), "pf" = 5
        const bool w4x = wf_is_4bit(h->wf) && (tiled == 32 || tiled == 64) && !split &&
                         (h->t_pf == 5 || h->t_pf == 6 || (h->t_pf == 7 && h->wf == LKM_W_INT4_B8 && !h->ps && !h->zp)) && (!h->zp || h->spu <= 2) &&
                         h->H % 128 == 0 && h->I % 128 == 0;
        if (w4x) pf = h->t_pf;                  // (6: with a loader wave per workgroup, gemm_w4e.h; "pd" = ring depth 3 / 4;
                                                //  7: two memory queues, gemm_w4s.h; "pd" = depth of the weight register ring)
        // round 5: uint4b8 (exact mode) on 64-row tiles takes the loader-wave kernel (gemm_w4e.h) BY DEFAULT for both GEMMs.
        // Round 4 left the choice to the first-call autotune; with the autotune off (the engine's default, and bench.py's since
        // round 5) the better plan must be the planner's.  Same box, captured step, alternating, after the consumer loop was
        // unrolled by the ring depth (profiles/r05_int4_unrolled_ab.log, Mixtral int4-g128 M=128, tile kernel -> gemm_w4e.h):
        // uniform step 243.0 -> 232.1 us (GEMM1 144.4 -> 137.3-139.7, GEMM2 79.4 -> 72.3-72.8), Zipf 253.2 -> 247.0-249.6 (GEMM1
        // 147.7 -> 147.0-148.7, GEMM2 88.3 -> 79.4-80.2).  Before the unroll GEMM1 was behind (profiles/r05_int4_default_ab.log).
        // (Eager per-kernel sweeps over-state such differences, profiles/r05_plan_robustness_sweep.log: plans are judged through
        // the graph.)  "pf" = -1 gives the tile kernel back.
        const bool w4e_dflt = h->wf == LKM_W_INT4_B8 && !h->ps && (!h->zp || h->spu <= 2) && tiled == 64 && !split && !g2_only && h->t_pf == 0 &&
                              h->H % 128 == 0 && h->I % 128 == 0;
        if (w4e_dflt) pf = 6;
        if (tiled == 256 && h->a8) {          // the fp8 x fp8 prefill kernels are the only 256-row variants of the format
            // 9 = gemm_prefill_a8w.h (weights straight to registers, tokens through a 4-stage LDS ring, equal token tiles);
            // round 2's kernel ("pf" = 8: both operands through two LDS buffers) was removed in round 4 -- K loops outside
            // the 8..64 units the item-boundary pipeline needs keep the 128-row tiles (the rule that picks 256 above)
            pf = 9;
            waves = 8;
            // The XCD-aware runs (dispatch.hip): for the round-2 kernel a knob ("xcd" = 1) -- they cut GEMM1's L2-miss
            // traffic 5.8 -> 3.55 GB and that kernel ran 4-6 % SLOWER (bound by the round trip of one 66 KiB DMA burst
            // per CU, profiles/r02_glm_a8_prefill.md).  The round-3 kernel keeps ~190 KiB in flight per CU and IS
            // sensitive to where its bytes come from: GEMM1 1127 -> 982 us, GEMM2 762 -> 653 us with the runs
            // (profiles/r03_a8w_*.log), so they are its default ("xcd" = -1 switches them off).
            if (pf == 9 && h->t_xcd >= 0) pl->xcd1 = pl->xcd2 = 1;
        }
        // 16-bit weights, 256-row tiles, many experts (GLM-4.5-Air prefill: 128 experts x ~2 tiles): the XCD-aware runs keep
        // the workgroups that share a weight panel on one L2 -- round 1 measured +-0 with equal ITEM counts per XCD; with the
        // runs cut by routed rows (round 3) the bf16 prefill step goes 3095 -> 2853 us uniform, 3158 -> 3006 us Zipf
        // (profiles/r04_prefill_plan_sweep.log).  Few large experts (Mixtral) lose with it and keep the plain grid.
        if (tiled == 256 && (w16 || w8a16 || w4pf) && n_act >= 32 && h->t_xcd >= 0 && (!pf || pf == 8))
            pl->xcd1 = pl->xcd2 = 1;
        // gemm_w4e.h reads "pd" as its LDS ring depth: 3 slots (two workgroups per CU), or 2 -- round 6: a slot is refilled right
        // behind its barrier, 51 KiB instead of 77 per workgroup, three workgroups per CU.  Same box, captured step, Mixtral
        // int4 M=128 (profiles/r06_w4e_s2.log): GEMM1 132.4 -> 130.4 us with two slots, GEMM2 70.1 -> 77.8 (its K loop of 28 units
        // per slab is all start-up: the deeper ring matters) -- so GEMM1 takes two, GEMM2 three
        if (pf == 6) {
            pd1 = tiled == 64 ? 2 : 3;
            pd2 = 3;
        }
        // ... and seven consumers per workgroup where the expert's row groups come in sevens (gemm_w4e.h launch_w4e_if: Mixtral's
        // 896 gate / up tile pairs = 128 x 7 -> 1024 workgroups = two exact rounds of the chip; GEMM1 132.5 -> 124.3 us).  GEMM2
        // keeps four (its 128 groups x 4 slabs are 1024 workgroups already; seven measured 91 vs 71 us)
        int waves1 = waves;
        if (pf == 6 && tiled == 64 && h->wf == LKM_W_INT4_B8 && h->t_waves == 0 && (h->gated ? h->T1_half : h->T1_half / 2) % 7 == 0)
            waves1 = 7;
        if (h->t_pd1 > 0) pd1 = h->t_pd1;
        if (h->t_pd2 > 0) pd2 = h->t_pd2;
        // tiled GEMM2 split-K: few experts per rank (expert parallel) leave T2/waves workgroups per token
        // tile -- Mixtral EP=8: 64 of them; slabs are summed by combine_kernel.  Not with the hybrid plan
        // (the skinny GEMM2 shares the slab layout and runs sk = 1 there) nor with the prefill kernel.
        int sk2 = 1;
        if (!split && (!pf || pf == 5 || pf == 6 || pf == 7)) {
            const int tpw = pf >= 5 ? 2 : nt2;     // weight tiles per wave
            const long long wg = (long long)n_act * ((avg_rows + tiled - 1) / tiled) * ((h->T2 + waves * tpw - 1) / (waves * tpw));
            // 4-bit weights: a workgroup's K loop is latency-bound, so more and shorter ones pay up to 4 slabs
            // (Mixtral M=128, 512 workgroups at sk 1: GEMM2 int4 107 -> 95 us, NVFP4 100 -> 86, MXFP4 88 -> 75);
            // 16-bit and fp8 lose with any split once the grid covers the chip (bf16 154 -> 163, fp8 90 -> 95)
            const long long wg_target = wf_is_4bit(h->wf) ? 2048 : 512;
            const int sk_cap = wf_is_4bit(h->wf) ? 4 : 8;
            while (sk2 < sk_cap && wg * sk2 < wg_target && h->U2 / (sk2 * 2) >= 4) sk2 *= 2;
            const size_t y_rows = h->arena->y_elems / h->H;
            while (sk2 > 1 && (size_t)sk2 * n_slots > y_rows) sk2 /= 2;
            if (h->t_sk2 > 0) sk2 = h->t_sk2;
        }
        // fp8 W8A16 at prefill sizes (MOE_FP8.gpu_prefill): GEMM2 on eight waves per workgroup -- GLM-4.5-Air M=8192
        // GEMM2 1474 -> 1156 us uniform, 1491 -> 1200 Zipf (GEMM1 unchanged by it): profiles/r04_prefill_plan_sweep.log
        int waves2 = waves;
        if (h->wf == LKM_W_FP8_E4M3 && !h->a8 && tiled == 64 && avg_rows >= 192 && h->t_waves == 0 && nt2 == 1) waves2 = 8;
        // gemm_w4e.h with seven consumers: "kw1" = 2 -> two sets of seven row groups per workgroup, one stream (LaunchCfg::kw)
        // (measured, profiles/r06_w4e_two_sets_ab.log: 512 workgroups, all resident at once -- GEMM1 126.8 -> 128.9 us uniform,
        // 142.7 -> 161.7 under Zipf: the bigger static items cost more under skew than the spared start-ups give back; opt-in)
        const int sets1 = (pf == 6 && waves1 == 7 && h->t_kw1 == 2) ? 2 : 1;
        pl->t1 = LaunchCfg{nt1, tiled / 16, sets1, 1, tiled, waves1, pd1, pf};
        pl->t2 = LaunchCfg{nt2, tiled / 16, 1, sk2, tiled, waves2, pd2, pf};
        if (g2_only) pl->t1 = LaunchCfg{0, 0, 0, 0, 0, 0, 0, 0};
        // Mixed tile heights (round-4 verdict item 6), OPT-IN: "mixed" = n > 0.  Decode batches of many-expert layers plan
        // 32- or 64-row tiles for the row count an expert is LIKELY to get; a skewed router hands one expert several times that
        // (DeepSeek-V3 rank slice under Zipf: 18 of 32 experts hit, the hot one 126 of 256 rows = four 32-row tiles = four
        // passes over its weights).  With the knob, experts with more than n rows take 128-row tiles (one pass; the
        // gemm_tiled_kernel<..., 8 waves> variant every format has), decided ON THE DEVICE by the sort, two launches per GEMM.
        // Measured (profiles/r05_mixed_heights_ab.log) and therefore NOT the default: the GEMM intervals of the profiled
        // call drop (DSv3 256 experts, Zipf: GEMM1 403 -> 274-308 us, GEMM2 215 -> 144-169) but the captured step does not
        // (675 -> 649-723 us; rank slice 173 -> 293): a 128-row tile is 16 workgroups per expert, each pulling 1.8 MB
        // alone -- latency-bound from HBM in the step, cache-resident under the profiler's eight back-to-back launches (one
        // expert = 29 MB stays in the 256 MB Infinity Cache).  Four 32-row tiles are 128 workgroups that share the panel
        // through the L2: more bytes from the L2, fewer from a latency-bound stream.  What would pay is the big tiles
        // running BESIDE the small ones (one launch), not before them.
        // 16-bit and fp8 weights on gemm_tiled_kernel only, single-workgroup sorts only (<= 4096 slots).
        if ((tiled == 32 || tiled == 64) && !split && !g2_only && !pf && !wf_is_4bit(h->wf) && n_slots <= 4096 &&
            h->t_mixed > 0 && h->t_xcd <= 0) {
            pl->big_rows = 128;
            pl->big_min = h->t_mixed;
            pl->b1 = LaunchCfg{1, 8, 1, 1, 128, 8, 2, 0};
            pl->b2 = LaunchCfg{1, 8, 1, sk2, 128, 8, 2, 0};
        }
        if (!split && !g2_only) return;
    }
    // ---- skinny geometry (only register-resident variants exist: gated needs nt<=2 and nt*tb<=4;
    // otherwise nt*tb<=8)
    int nt1 = 1;
    if (tb >= 4 && !h->gated && (long long)n_act * (h->T1_half / 2) >= 2 * kMinWaves) nt1 = 2;
    // 4-bit weights: a tile-unit is only 1 KiB, two tiles per wave double the bytes in flight per wave
    // and halve the token-operand loads per weight byte (MXFP4 M=32: gemm1 110 -> 88 us)
    if (wf_is_4bit(h->wf) && tb == 2 && (long long)n_act * (h->T1_half / 2) >= kMinWaves) nt1 = 2;
    if (h->t_nt1 > 0) nt1 = h->t_nt1;
    int kw = 1;
    {
        long long waves = (long long)n_act * (h->T1_half / nt1);
        while (kw < 8 && waves * kw < kMinWaves && h->U1 / (kw * 2) >= 2) kw *= 2;
    }
    // 16-bit weights, waves that may hold two token blocks (a hot expert under skewed routing does): four waves per
    // workgroup split K and reduce through LDS -- a quarter of the wave length, so the slow two-block waves of the hot
    // expert stop setting the tail.  Mixtral bf16 M=32 through bench.py (graph, alternating): uniform 440.6 -> 439.4 us,
    // Zipf 455.9 -> 448.6-449.7 (kw 2: 452; kw 8: 467-470): profiles/r03_streamer_kw_ab.log
    // (fp8 x fp8, Mixtral M=32: GEMM1 153.3 -> 148.7 us uniform, 157.7 -> 154.1 Zipf)
    if ((h->wf == LKM_W_BF16 || h->wf == LKM_W_F16 || h->a8) && h->gated && tb >= 2 && kw < 4 && h->U1 >= 32) kw = 4;
    if (h->t_kw1 > 0) kw = h->t_kw1;
    pl->s1 = LaunchCfg{nt1, tb, kw, 1, 0, 0, 0, 0};
    // sub-16-bit weights: two tiles per wave halve the token-operand loads per weight byte
    // (int4 M=32: gemm2 120 us -> 85 us)
    int nt2 = (wf_is_4bit(h->wf) || h->wf == LKM_W_FP8_E4M3) && tb >= 2 ? 2 : 1;
    if (tb >= 4 && (long long)n_act * (h->T2 / 2) >= 2 * kMinWaves) nt2 = 2;
    if (h->t_nt2 > 0) nt2 = h->t_nt2;
    int sk = 1;
    {
        long long waves = (long long)n_act * (h->T2 / nt2);
        // fp8 x fp8 with two tiles per wave: 1024 waves of the 164-register kernel (three per SIMD) already cover the
        // chip; splitting K only adds slab traffic (Mixtral fp8-W8A8 M=32: GEMM2 78.6 -> 75.0 us at sk 1)
        const long long min_waves = h->a8 ? kMinWaves / 2 : kMinWaves;
        while (sk < 8 && waves * sk < min_waves && h->U2 / (sk * 2) >= 2) sk *= 2;
    }
    if (h->t_sk2 > 0) sk = h->t_sk2;
    if (split) sk = 1;   // the tiled GEMM2 writes slab 0 only
    // split-K slabs must fit the partial buffer
    const size_t y_rows = h->arena->y_elems / h->H;
    while (sk > 1 && (size_t)sk * n_slots > y_rows) sk /= 2;
    pl->s2 = LaunchCfg{nt2, tb, 1, sk, 0, 0, 0, 0};
    if (g2_only) pl->s2 = LaunchCfg{0, 0, 0, 0, 0, 0, 0, 0};
}

// the no-scatter decode path (run_chunk): at most four tokens, streamer geometry of one tile / one token block per wave
constexpr int kDirectMaxTokens = 4;      // (tuning key "direct": -1 off, n > 0 = at most n tokens, up to 16)
static int direct_max_tokens(const LkmEngine* h) {
    return h->t_direct < 0 ? 0 : (h->t_direct > 0 ? (h->t_direct < 16 ? h->t_direct : 16) : kDirectMaxTokens);
}
static bool direct_plan(const LkmEngine* h, int M, int K, const Plan& pl) {
    if (!(M >= 1 && M <= direct_max_tokens(h) && K <= 16 && pl.s1.tb == 1 && pl.s1.nt == 1 && !pl.t1.tiled && !pl.t2.tiled))
        return false;
    if (M == 1 || h->t_direct > 0) return true;      // (an explicit token limit is taken at its word: experiments)
    // Without the scatter every SLOT streams its expert: two tokens that pick the same expert read it twice.  Expected
    // repeats across tokens ~ C(M,2) K^2 / E, each costing one expert's bytes at ~6.5 TB/s, against the ~6 us the two
    // saved launches are worth.  Many small experts (Qwen3-30B-A3B: 9.4 MB each, M <= 4; measured 35.6 vs 45.4 us at
    // M = 2, 55 vs 66 at M = 4) take the path, few large ones (Mixtral: 352 MB each; M = 3: 319 vs 226 us) do not.
    const double repeats = 0.5 * M * (M - 1) * (double)K * K / (double)h->E;
    const double expert_us = (double)h->weight_bytes / (double)h->E / 6.5e6;
    return repeats * expert_us <= 6.0;
}

// strides (elements) of the three input arrays and the id offset of lkm_forward_strided
struct InLayout {
    int64_t x_ld, ids_ld, tw_ld;
    int id_off;
    const RouteArgs* route = nullptr;   // lkm_forward_routed: the router runs inside the sort launch
};

// one chunk: rows [0,M) of the given pointers
static int run_chunk(LkmEngine* h, hipStream_t st, int M, int K, const void* x, const int32_t* ids,
                     const float* tw, void* out, int out_dt, const InLayout& il) {
    Arena* a = h->arena;
    const size_t n_slots = (size_t)M * K;
    Plan pl;
    pick_cfg(h, M, n_slots, &pl);
    const bool prof = h->prof;
    // profiling only: each GEMM is launched `rep` times back to back between its two events and the
    // interval divided by rep (lkm_get_profile).  One launch between two events also times the event
    // packets themselves (~10 us on MI355X: Mixtral GEMM1 288 us against the 274 us rocprofv3 reports
    // for the kernel); the repeats are idempotent -- same inputs, same outputs.
    const int rep = prof && h->t_prof_rep > 1 ? h->t_prof_rep : 1;
    h->prof_rep_used = rep;
    if (prof) {
        h->prof_stream = st;
        LKM_HIP_CHECK(hipEventRecord(h->ev[0], st));
    }
    int rc = LKM_OK;
    const int kb1 = ceil_div(h->H, 128), kb2 = ceil_div(h->I, 128);
    if (h->a8) {   // dynamic 1x128 fp8 quantisation of the token rows (once per token, not per slot)
        LKM_REQUIRE((size_t)M * h->H <= a->xq_n && n_slots * h->I <= a->aq_n, "fp8 activation scratch too small");
        rc = launch_quant_fp8_rows(st, x, (int)il.x_ld, h->adt, M, h->H, a->xq, a->xqs);
        if (rc != LKM_OK) return rc;
    }
    if (h->ps) {   // int4 fast mode: the activation-side term, one fp32 sum per (token, 128-k block)
        LKM_REQUIRE((size_t)M * kb1 <= a->xqs_n && n_slots * kb2 <= a->aqs_n, "int4 activation-sum scratch too small");
        rc = launch_rowsum128_rows(st, x, (int)il.x_ld, h->adt, M, h->H, a->xqs);
        if (rc != LKM_OK) return rc;
    }
    const int tile_rows = pl.t1.tiled ? pl.t1.tiled : pl.t2.tiled;
    // the round-3 fp8 prefill kernel takes token tiles of any height up to 256 in 32-row steps: the sort deals an
    // expert's rows to its tiles evenly (dispatch.hip tile_first_row)
    const int tile_rows_sort = (pl.t1.pf == 9 && pl.t2.pf == 9) ? pack_tile_rows(tile_rows, 32) : tile_rows;
    // Decode of one to four tokens: no scatter -- GEMM1 reads the router's ids itself (one workgroup row per SLOT, the
    // token's row as B operand) and GEMM2 multiplies the K slots of a token and forms the weighted sum in one workgroup
    // (grid row = token): two launches instead of four.  (Qwen3-30B-A3B M=1: 38 -> 32 us: launch latency, not bandwidth.
    // Two tokens that pick the same expert stream its weights twice -- concurrently, out of L2 -- which is why this stops
    // at four tokens.)
    int sk_direct = 1;
    if (M <= direct_max_tokens(h) && K <= 16) {
        const long long waves = (long long)n_slots * (h->T2 / 1);
        while (sk_direct * 2 * K <= 16 && waves * sk_direct < 2048 && h->U2 / (sk_direct * 2) >= 2) sk_direct *= 2;
    }
    const bool direct = direct_plan(h, M, K, pl) && il.ids_ld == K && il.tw_ld == K;
    // hybrid: experts with more than split_rows rows go to the tile list; mixed heights: the packed pair (pack_mixed_tiles)
    const int tile_min_sort = pl.big_rows ? pack_mixed_tiles(pl.big_rows, pl.big_min) : pl.split_rows;
    // big tiles of a mixed plan: experts with more than big_min rows, ceil(rows / big_rows) tiles each
    const int max_big_tiles = pl.big_rows ? (int)(n_slots / (size_t)pl.big_rows + n_slots / (size_t)(pl.big_min + 1)) + 1 : 0;
    const int max_active = (int)((size_t)h->E < n_slots ? (size_t)h->E : n_slots);
    const int max_tiles = tile_rows ? (int)(n_slots / tile_rows) + max_active : 0;
    // XCD-aware mapping: the sort kernel cuts the tile list into 8 runs of equal routed rows, none longer than
    // xcd_cap tiles (twice the mean: the 1-D grid is sized by it)
    const bool want_xcd = tile_rows && max_tiles <= 4096 && (h->t_xcd > 0 || pl.xcd1 || pl.xcd2);
    const int xcd_cap = want_xcd ? 2 * ((max_tiles + 7) / 8) + 1 : 0;
    if (!direct && il.route) {
        LKM_REQUIRE(launch_route_sort_ok(M, K, il.route->E, il.route->n_group, h->E), "forward_routed: step planned off both fused paths");
        rc = launch_route_sort(st, *il.route, il.id_off, h->E, a->counts, a->offsets, a->sorted_slot, a->pos_of_slot,
                               a->active, a->meta, tile_rows_sort, tile_min_sort, a->tile_e, a->tile_r0, xcd_cap);
        if (rc != LKM_OK) return rc;
    } else if (!direct) {
        rc = launch_sort(st, ids, K, (int)il.ids_ld, il.id_off, (int)n_slots, h->E, a->counts, a->offsets, a->sorted_slot, a->pos_of_slot,
                         a->active, a->meta, tile_rows_sort, tile_min_sort, a->tile_e, a->tile_r0, a->hist,
                         a->hist_cap, xcd_cap);
        if (rc != LKM_OK) return rc;
    }
    const int32_t *items1 = nullptr, *items2 = nullptr;
    if (pl.t1.pf == 9 && pl.t2.pf == 9 && !direct) {
        // the round-3 fp8 x fp8 prefill kernel walks its own item lists (dispatch.hip build_items_kernel)
        const int rg1 = a8w_row_groups(h, 1), rg2 = a8w_row_groups(h, 2);
        LKM_REQUIRE((size_t)max_tiles * (size_t)(rg1 + rg2) * 4 <= a->items_cap, "prefill item lists: %d tiles exceed the arena", max_tiles);
        items1 = a->items;
        items2 = a->items + (size_t)max_tiles * rg1 * 4;
        rc = launch_build_items(st, a->tile_e, a->tile_r0, a->counts, a->offsets, a->meta,
                                xcd_cap && (h->t_xcd > 0 || pl.xcd1) ? 1 : 0, rg1, rg2, max_tiles, a->items, a->items + (size_t)max_tiles * rg1 * 4);
        if (rc != LKM_OK) return rc;
    }
    if (prof) LKM_HIP_CHECK(hipEventRecord(h->ev[1], st));
    // weights are read once per step when an expert's rows fit one token tile
    const int stream_nt = tile_rows && !pl.split_rows
                              ? ((n_slots / (size_t)(max_active > 0 ? max_active : 1)) <= (size_t)tile_rows)
                              : 1;

    GemmParams p1{};
    p1.w = h->w13;
    p1.s = h->s13;
    p1.spu = h->spu;
    set_w_layout(p1, h->T1_half * (h->gated ? 2 : 1), h->U1, h->loads);
    p1.gs = h->gs13;
    p1.xcd_map = (h->t_xcd > 0 || pl.xcd1) ? xcd_cap : 0;
    p1.tile_uniform_scale = h->cfg.groupN > 0 && h->cfg.groupN % 16 == 0;
    p1.dbg = h->t_dbg;
    p1.x_rows = M;
    p1.T_half = h->T1_half;
    p1.halves = h->gated ? 2 : 1;
    p1.U = h->U1;
    p1.Kreal = h->H;
    p1.n_real = h->I;
    p1.x = h->a8 ? (const void*)a->xq : x;
    p1.ldx = h->a8 ? h->H : (int)il.x_ld;
    p1.xscale = a->xqs;
    p1.ld_xscale = kb1;
    p1.round_gemm1 = h->a8 ? 1 : 0;
    p1.top_k = K;
    p1.rcp_top_k = 1.0f / (float)K;
    // The gated GEMM1 of the fp8 prefill kernel quantises its own output (the 1 x 128 groups of the W8A8 intermediate:
    // one item = 256 rows x one group) where every 128-column group is whole: no 16-bit intermediate, no quantiser launch.
    // Tuning key "fuseq" = -1 keeps the separate pass (same bytes: tests/test_gpu_moe.py).
    const bool fuse_q = h->a8 && h->gated && !direct && pl.t1.tiled == 256 && pl.t1.pf == 9 && !pl.s1.tb && !pl.split_rows &&      // (every row of the intermediate comes from that kernel)
                         h->I % 128 == 0 && h->ld_act % 8 == 0 &&
                        h->cfg.activation_type != LKM_ACT_SWIGLUOAI && h->t_fuseq >= 0 && !(h->t_dbg & 0x3fb);
    if (fuse_q) {
        p1.out_q = (unsigned char*)a->aq;
        p1.out_qs = a->aqs;
        p1.ld_qs = kb2;
    }
    p1.counts = a->counts;
    p1.offsets = a->offsets;
    p1.active = a->active;
    p1.meta = a->meta;
    p1.sorted_slot = a->sorted_slot;
    p1.tile_e = a->tile_e;
    p1.tile_r0 = a->tile_r0;
    p1.items = items1;
    p1.max_rows = pl.split_rows;
    p1.stream_nt = stream_nt;
    p1.out = a->act;
    p1.ldo = h->ld_act;
    p1.sk_stride = 0;
    p1.SK = 1;
    p1.act_type = h->cfg.activation_type;
    p1.alpha = h->cfg.swiglu_alpha;
    p1.limit = h->cfg.swiglu_limit;
    if (direct) {
        p1.direct_ids = ids;
        p1.direct_w = tw;
        p1.direct_E = h->E;
        p1.direct_id_off = il.id_off;
        if (il.route) {          // lkm_forward_routed at M = 1: GEMM1 routes the row itself (ids / tw = its outputs)
            p1.route_on = 1;
            p1.route = *il.route;
        }
    }
    note_phase(1);
    for (int r = 0; r < rep; ++r) {
        if (pl.s1.tb) {
            p1.groups = h->T1_half / pl.s1.nt;
            rc = launch_gemm1(st, h->wfk, h->adt, pl.s1, p1, h->gated, direct ? (int)n_slots : max_active);
            if (rc != LKM_OK) return rc;
        }
        if (pl.big_rows) {           // the big tiles first (the longest workgroups), then the rest of the list
            GemmParams pb = p1;
            pb.tile_hi_meta = 4;
            rc = launch_gemm1_tiled(st, h->wfk, h->adt, pl.b1, pb, h->gated, max_big_tiles);
            if (rc != LKM_OK) return rc;
            p1.tile_lo_meta = 4;
        }
        if (pl.t1.tiled) {
            rc = launch_gemm1_tiled(st, h->wfk, h->adt, pl.t1, p1, h->gated, max_tiles);
            if (rc != LKM_OK) return rc;
        }
    }
    g_note_phase = 0;
    if (prof) LKM_HIP_CHECK(hipEventRecord(h->ev[2], st));

    GemmParams p2{};
    p2.w = h->w2;
    p2.s = h->s2;
    p2.spu = h->spu;
    set_w_layout(p2, h->T2, h->U2, h->loads);
    p2.gs = h->gs2;
    p2.xcd_map = (h->t_xcd > 0 || pl.xcd2) ? xcd_cap : 0;
    p2.tile_uniform_scale = p1.tile_uniform_scale;
    p2.dbg = h->t_dbg;
    p2.x_rows = (long long)n_slots;
    p2.T_half = h->T2;
    p2.halves = 1;
    p2.U = h->U2;
    p2.Kreal = h->I;
    p2.n_real = h->H;
    if (h->a8 && !fuse_q) {   // re-quantise the intermediate (per_token_group_quant of act_out, test_block_fp8.py:128)
        rc = launch_quant_fp8_rows(st, a->act, h->ld_act, h->adt, (int)n_slots, h->I, a->aq, a->aqs);
        if (rc != LKM_OK) return rc;
    }
    if (h->ps) {
        rc = launch_rowsum128_rows(st, a->act, h->ld_act, h->adt, (int)n_slots, h->I, a->aqs);
        if (rc != LKM_OK) return rc;
    }
    p2.x = h->a8 ? (const void*)a->aq : a->act;
    p2.ldx = h->ld_act;
    p2.xscale = a->aqs;
    p2.ld_xscale = kb2;
    p2.top_k = K;
    p2.rcp_top_k = 1.0f / (float)K;
    p2.counts = a->counts;
    p2.offsets = a->offsets;
    p2.active = a->active;
    p2.meta = a->meta;
    p2.sorted_slot = a->sorted_slot;
    p2.tile_e = a->tile_e;
    p2.tile_r0 = a->tile_r0;
    p2.items = items2;
    p2.max_rows = pl.split_rows;
    p2.stream_nt = stream_nt;
    p2.out = a->y;
    p2.ldo = h->H;
    p2.sk_stride = n_slots * (size_t)h->H;
    const int sk = pl.s2.tb ? pl.s2.sk : (pl.t2.tiled ? pl.t2.sk : 1);
    p2.SK = sk;
    // block-fp8 W8A8 on the prefill kernel: GEMM2 rounds its output to the activation dtype like the reference's
    // native_w8a8_block_matmul (output_dtype) -- half the bytes written here and read back by the combine
    // 16-bit weights on gemm_prefill.h ("pf" = 8, prefill sizes only): the same -- what the in-tree GPU operator's GEMM2 does
    // (fused_moe.py writes intermediate_cache3 in the hidden dtype before moe_sum); GLM-4.5-Air bf16 M=8192: GEMM2 815 ->
    // 728 us, combine 248 -> 136.  "ydt" = -1 keeps the fp32 partials of the smaller-batch kernels.
    const bool w16_pf = (h->wf == LKM_W_BF16 || h->wf == LKM_W_F16 || (h->wf == LKM_W_FP8_E4M3 && !h->a8) || wf_is_4bit(h->wf)) && pl.t2.tiled == 256 &&
                        pl.t2.pf == 8 && h->t_ydt >= 0;
    const int y_dt = (((h->a8 && pl.t2.pf >= 8) || w16_pf) && pl.t2.tiled == 256 && !pl.s2.tb && sk == 1) ? h->adt : LKM_DT_F32;
    p2.y_dt = y_dt;
    if (direct) {
        p2.direct_ids = ids;
        p2.direct_w = tw;
        p2.direct_E = h->E;
        p2.direct_id_off = il.id_off;
        p2.direct_out_dt = out_dt;
        p2.out = out;
        p2.SK = sk_direct;
        LaunchCfg dc = pl.s2;
        dc.nt = 1;
        note_phase(2);
        for (int r = 0; r < rep; ++r) {
            rc = launch_gemm2_direct(st, h->wfk, h->adt, dc, p2, K);
            if (rc != LKM_OK) return rc;
        }
        g_note_phase = 0;
        memcpy(h->last_fn, g_note_fn, sizeof(h->last_fn));
        if (prof) {
            LKM_HIP_CHECK(hipEventRecord(h->ev[3], st));
            LKM_HIP_CHECK(hipEventRecord(h->ev[4], st));
            h->prof_valid = true;
        }
        snprintf(h->last_desc, sizeof(h->last_desc),
                 "M=%d K=%d | direct (no sort / combine launch) | skinny g1 nt=1 tb=1 kw=%d, g2+combine nt=1 sk=%d | nt_loads=1",
                 M, K, pl.s1.kw, sk_direct);
        return LKM_OK;
    }
    // (GEMM2 + top-k combine in one launch for few active experts -- round 2's gemm2_combine_kernel, tuning key "fuse" = 1
    // -- was bit-identical and 2.7 us SLOWER than the two launches (profiles/r02_decode_fusion.md); removed in round 4)
    note_phase(2);
    for (int r = 0; r < rep; ++r) {
        if (pl.s2.tb) {
            p2.groups = h->T2 / pl.s2.nt;
            rc = launch_gemm2(st, h->wfk, h->adt, pl.s2, p2, max_active);
            if (rc != LKM_OK) return rc;
        }
        if (pl.big_rows) {
            GemmParams pb = p2;
            pb.tile_hi_meta = 4;
            rc = launch_gemm2_tiled(st, h->wfk, h->adt, pl.b2, pb, max_big_tiles);
            if (rc != LKM_OK) return rc;
            p2.tile_lo_meta = 4;
        }
        if (pl.t2.tiled) {
            rc = launch_gemm2_tiled(st, h->wfk, h->adt, pl.t2, p2, max_tiles);
            if (rc != LKM_OK) return rc;
        }
    }
    g_note_phase = 0;
    memcpy(h->last_fn, g_note_fn, sizeof(h->last_fn));
    if (prof) LKM_HIP_CHECK(hipEventRecord(h->ev[3], st));

    rc = launch_combine(st, a->y, y_dt, sk, p2.sk_stride, a->pos_of_slot, tw, (int)il.tw_ld, M, K, h->H, out, out_dt);
    if (rc != LKM_OK) return rc;
    if (prof) {
        LKM_HIP_CHECK(hipEventRecord(h->ev[4], st));
        h->prof_valid = true;
    }
    snprintf(h->last_desc, sizeof(h->last_desc),
             "M=%d K=%d | %s%s | skinny g1 nt=%d tb=%d kw=%d, g2 nt=%d tb=%d sk=%d | tiled g1 nt=%d, g2 nt=%d, tm=%d waves=%d pd=%d/%d split=%d pf=%d xcd=%d/%d mixed=%d>%d | nt_loads=%d",
             M, K, pl.s1.tb ? "skinny" : "", pl.t1.tiled ? (pl.s1.tb ? "+tiled" : "tiled") : "", pl.s1.nt,
             pl.s1.tb, pl.s1.kw, pl.s2.nt, pl.s2.tb, pl.s2.sk, pl.t1.nt, pl.t2.nt, tile_rows, pl.t1.waves,
             pl.t1.pd, pl.t2.pd, pl.split_rows, pl.t1.pf, p1.xcd_map ? 1 : 0, p2.xcd_map ? 1 : 0, pl.big_rows, pl.big_min, stream_nt);
    return LKM_OK;
}

// tokens per chunk: the arena was sized with cfg.top_k; a different K only changes how many tokens fit a chunk
static size_t chunk_tokens(const LkmEngine* h, int K) {
    size_t chunk = h->arena->cap_slots / (size_t)K;
    if (h->arena->act_elems / ((size_t)K * h->ld_act) < chunk) chunk = h->arena->act_elems / ((size_t)K * h->ld_act);
    if (h->arena->y_elems / ((size_t)K * h->H) < chunk) chunk = h->arena->y_elems / ((size_t)K * h->H);
    if (h->a8) {
        if (h->arena->xq_n / (size_t)h->H < chunk) chunk = h->arena->xq_n / (size_t)h->H;
        if (h->arena->aq_n / ((size_t)K * h->I) < chunk) chunk = h->arena->aq_n / ((size_t)K * h->I);
    }
    if (h->a8 || h->ps) {
        const size_t kb1 = ceil_div(h->H, 128), kb2 = ceil_div(h->I, 128);
        if (h->arena->xqs_n / kb1 < chunk) chunk = h->arena->xqs_n / kb1;
        if (h->arena->aqs_n / ((size_t)K * kb2) < chunk) chunk = h->arena->aqs_n / ((size_t)K * kb2);
    }
    return chunk;
}

// ---- first-call micro-autotune of one chunk's plan (tuning key "autotune" = 1)
// The thresholds of pick_cfg were measured on three model shapes and two box classes; the 4-bit plans in particular
// swing 8-25 % between the classes.  With "autotune" on, the first eager call of a step shape (M <= 1024 tokens; not while
// the stream is capturing; not while an explicit plan is forced through "pf" / "tiled" / "pd*") runs each candidate once
// untimed and `kTuneReps` times between two events on the caller's inputs, keeps the fastest and then runs it once more so
// that `out` holds ITS result.  Candidates: the default plan, the streamer where the default uses tiles and vice versa,
// 32- / 64-row tiles, the deeper weight ring, and for the 4-bit formats the 32x32-MFMA kernels (gemm_w4e.h / gemm_w4x.h).  Every
// candidate is a plan the parity tests cover; which one wins depends on timing, so two processes may choose
// differently (results then differ in fp32 summation order, inside the stated tolerance) -- the reason it is opt-in.
constexpr int kTuneReps = 3;
static bool tune_stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}
// the step shape a plan is remembered for: token count, top-k, output dtype, fused router, expert-parallel share and
// whether the three input arrays are dense (the strides do not change the plan, dense / strided callers may still differ
// in M); every planning knob of lkm_set_tuning clears the table instead of widening the key
static uint64_t tune_key(const LkmEngine* h, int M, int K, int out_dt, const InLayout& il) {
    return (uint64_t)(unsigned)M | ((uint64_t)(K & 0xff) << 32) | ((uint64_t)(out_dt & 0xf) << 40) |
           ((uint64_t)(il.route ? 1 : 0) << 44) | ((uint64_t)(h->t_valid_den & 0xff) << 45) |
           ((uint64_t)(il.x_ld != h->H ? 1 : 0) << 53) | ((uint64_t)(il.ids_ld != K ? 1 : 0) << 54) |
           ((uint64_t)(il.tw_ld != K ? 1 : 0) << 55);
}
// candidate plans of a step shape: a function of the engine's configuration and the shape only, so every rank of an
// expert-parallel group with equal local shapes builds the same list (lkm_tuned_plan_set agrees on an index into it)
static std::vector<LkmEngine::TunedPlan> tune_candidates(const LkmEngine* h, int M, int K) {
    std::vector<LkmEngine::TunedPlan> cands;
    cands.push_back({0, 0, 0, 0, 0.f, "default"});
    Plan pl;
    pick_cfg(h, M, (size_t)M * K, &pl);
    const int def_tiled = pl.t1.tiled ? pl.t1.tiled : (pl.t2.tiled && !pl.s1.tb ? pl.t2.tiled : 0);
    if (def_tiled && def_tiled <= 64) cands.push_back({0, -1, 0, 0, 0.f, "streamer"});
    if (def_tiled != 64 && !(pl.t1.tiled > 64)) cands.push_back({0, 64, 0, 0, 0.f, "64-row tiles"});
    if (def_tiled != 32 && wf_is_4bit(h->wf) && M * K <= 64 * h->E) cands.push_back({0, 32, 0, 0, 0.f, "32-row tiles"});
    if (def_tiled && def_tiled <= 64 && pl.t1.pd == 2) cands.push_back({0, def_tiled, 4, 0, 0.f, "weight ring depth 4 (GEMM1)"});
    if (wf_is_4bit(h->wf) && !h->ps && !h->zp && h->H % 128 == 0 && h->I % 128 == 0 && M >= 32) {
        cands.push_back({6, def_tiled == 32 ? 32 : 64, 0, 0, 0.f, "32x32-MFMA + loader wave (gemm_w4e.h)"});
        cands.push_back({5, def_tiled == 32 ? 32 : 64, 0, 0, 0.f, "32x32-MFMA (gemm_w4x.h)"});
    }
    return cands;
}
namespace {
// forces a candidate's knobs on the engine for one scope: every return path (incl. LKM_HIP_CHECK's) restores them
struct ForcedPlan {
    LkmEngine* h;
    ForcedPlan(LkmEngine* h_, const LkmEngine::TunedPlan& c) : h(h_) { h->t_pf = c.pf; h->t_tiled = c.tiled; h->t_pd1 = c.pd1; h->t_pd2 = c.pd2; }
    ~ForcedPlan() { h->t_pf = h->t_tiled = h->t_pd1 = h->t_pd2 = 0; }
};
}  // namespace
static int time_candidate(LkmEngine* h, hipStream_t st, const LkmEngine::TunedPlan& c, int M, int K, const void* x,
                          const int32_t* ids, const float* tw, void* out, int out_dt, const InLayout& il, float* us) {
    ForcedPlan fp(h, c);
    *us = 1e30f;
    int rc = run_chunk(h, st, M, K, x, ids, tw, out, out_dt, il);       // untimed (first-touch, code load)
    if (rc != LKM_OK) return rc;
    LKM_HIP_CHECK(hipEventRecord(h->tune_ev[0], st));
    for (int r = 0; r < kTuneReps && rc == LKM_OK; ++r) rc = run_chunk(h, st, M, K, x, ids, tw, out, out_dt, il);
    LKM_HIP_CHECK(hipEventRecord(h->tune_ev[1], st));
    if (rc != LKM_OK) return rc;
    LKM_HIP_CHECK(hipEventSynchronize(h->tune_ev[1]));
    float ms = 0.f;
    LKM_HIP_CHECK(hipEventElapsedTime(&ms, h->tune_ev[0], h->tune_ev[1]));
    *us = ms * 1e3f / kTuneReps;
    return LKM_OK;
}
static int run_chunk_tuned(LkmEngine* h, hipStream_t st, int M, int K, const void* x, const int32_t* ids,
                           const float* tw, void* out, int out_dt, const InLayout& il) {
    const bool forced = h->t_pf != 0 || h->t_tiled != 0 || h->t_pd1 != 0 || h->t_pd2 != 0 || h->t_waves != 0;
    if (!h->t_autotune || forced || M < 8 || M > 1024) return run_chunk(h, st, M, K, x, ids, tw, out, out_dt, il);
    const uint64_t key = tune_key(h, M, K, out_dt, il);
    auto it = h->tuned.find(key);
    if (it == h->tuned.end()) {
        // (a profiled call times one plan's kernels: it takes the remembered plan but never starts a tuning pass;
        //  an in-place caller -- `out` overlapping the token rows -- cannot be run repeatedly on its own buffers)
        const char *xb = (const char*)x, *xe = xb + (size_t)M * (size_t)il.x_ld * 2;
        const char *ob = (const char*)out, *oe = ob + (size_t)M * (size_t)h->H * (out_dt == LKM_DT_F32 ? 4 : 2);
        const bool aliased = xb < oe && ob < xe;
        if (h->prof || aliased || tune_stream_capturing(st) || h->tuned.size() >= 256)
            return run_chunk(h, st, M, K, x, ids, tw, out, out_dt, il);
        std::vector<LkmEngine::TunedPlan> cands = tune_candidates(h, M, K);
        if (!h->tune_ev[0]) {
            LKM_HIP_CHECK(hipEventCreate(&h->tune_ev[0]));
            LKM_HIP_CHECK(hipEventCreate(&h->tune_ev[1]));
        }
        int best = 0;
        for (size_t c = 0; c < cands.size(); ++c) {
            // (a candidate the shape does not admit is not an error of the step; a HIP error is)
            const int rc = time_candidate(h, st, cands[c], M, K, x, ids, tw, out, out_dt, il, &cands[c].us);
            if (rc == LKM_E_HIP) return rc;
            if (cands[c].us < cands[best].us) best = (int)c;
        }
        LKM_REQUIRE(cands[best].us < 1e29f, "autotune: no candidate plan ran");
        cands[best].index = best;
        it = h->tuned.emplace(key, cands[best]).first;
        h->tuned_shape[key] = {M, K};
    }
    int rc;
    {
        ForcedPlan fp(h, it->second);
        rc = run_chunk(h, st, M, K, x, ids, tw, out, out_dt, il);
    }
    if (rc == LKM_OK) {
        const size_t n = strlen(h->last_desc);
        snprintf(h->last_desc + n, sizeof(h->last_desc) - n, " | autotuned: %s (%.1f us)", it->second.what, it->second.us);
    }
    return rc;
}

// copies between the caller's rows of Hu elements and the engine's aligned scratch rows of H = round_up(Hu, 8)
template <typename T>
__global__ __launch_bounds__(256) void pad_rows_kernel(const T* __restrict__ src, long long ld_src, T* __restrict__ dst,
                                                       long long ld_dst, int rows, int cols_src, int cols_dst) {
    const long long n = (long long)rows * cols_dst;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / cols_dst;
        const int c = (int)(i - r * cols_dst);
        dst[r * ld_dst + c] = c < cols_src ? src[r * ld_src + c] : T(0);
    }
}
template <typename T>
static void launch_pad_rows(hipStream_t st, const void* src, long long ld_src, void* dst, long long ld_dst, int rows,
                            int cols_src, int cols_dst) {
    const long long n = (long long)rows * cols_dst;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(pad_rows_kernel<T>, dim3(blocks ? blocks : 1), dim3(256), 0, st, (const T*)src, ld_src, (T*)dst, ld_dst,
                       rows, cols_src, cols_dst);
}

static int run_device(LkmHandle h, hipStream_t st, int M, int K, const void* x, const int32_t* ids,
                      const float* tw, void* out, int out_dt, const InLayout* layout = nullptr) {
    LKM_REQUIRE(h, "null engine handle");
    LKM_REQUIRE(M >= 0, "num_tokens=%d < 0", M);
    LKM_REQUIRE(K > 0 && K <= 64, "top_k=%d out of range", K);
    if (M == 0) return LKM_OK;
    LKM_REQUIRE(x && ids && tw && out, "null device pointer");
    const bool odd = h->Hu != h->H;     // hidden_size % 8 != 0: rows travel through the aligned scratch matrices
    LKM_REQUIRE(odd || (((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0), "hidden/out pointers must be 16-byte aligned");
    LKM_HIP_CHECK(hipSetDevice(h->device));
    InLayout il = layout ? *layout : InLayout{h->Hu, K, K, 0};
    LKM_REQUIRE(il.x_ld >= h->Hu && (odd || il.x_ld % 8 == 0) && il.x_ld < ((int64_t)1 << 31), "hidden row stride %lld must be >= H, a multiple of 8 and < 2^31", (long long)il.x_ld);
    LKM_REQUIRE(il.ids_ld >= K && il.tw_ld >= K && il.ids_ld < ((int64_t)1 << 31) && il.tw_ld < ((int64_t)1 << 31), "ids / weights row strides must be >= top_k");
    size_t chunk = chunk_tokens(h, K);
    if (odd && h->pad_tokens < chunk) chunk = h->pad_tokens;
    LKM_REQUIRE(chunk > 0, "scratch arena too small for top_k=%d", K);
    const size_t osz = out_dt == LKM_DT_F32 ? 4 : 2;
    const size_t xrow = (size_t)il.x_ld * 2, orow = (size_t)h->Hu * osz;
    const int64_t x_ld_user = il.x_ld;
    if (odd) il.x_ld = h->H;
    LKM_REQUIRE(!il.route || chunk >= (size_t)M, "forward_routed: a fused router needs the batch in one chunk (M=%d, chunk=%zu)", M, chunk);
    for (size_t m0 = 0; m0 < (size_t)M; m0 += chunk) {
        const int mc = (int)((size_t)M - m0 < chunk ? (size_t)M - m0 : chunk);
        const void* xc = (const char*)x + m0 * xrow;
        void* oc = (char*)out + m0 * orow;
        if (odd) {
            launch_pad_rows<unsigned short>(st, xc, x_ld_user, h->pad_x, h->H, mc, h->Hu, h->H);
            xc = h->pad_x;
            oc = h->pad_o;
        }
        int rc = run_chunk_tuned(h, st, mc, K, xc, ids + m0 * il.ids_ld, tw + m0 * il.tw_ld, oc, out_dt, il);
        if (rc != LKM_OK) return rc;
        if (odd) {
            if (osz == 4) launch_pad_rows<float>(st, h->pad_o, h->H, (char*)out + m0 * orow, h->Hu, mc, h->Hu, h->Hu);
            else launch_pad_rows<unsigned short>(st, h->pad_o, h->H, (char*)out + m0 * orow, h->Hu, mc, h->Hu, h->Hu);
            LKM_HIP_CHECK(hipGetLastError());
        }
    }
    return LKM_OK;
}

extern "C" int lkm_decode(LkmHandle h, void* stream, int32_t num_tokens, int32_t top_k,
                          const void* hidden, const int32_t* topk_ids, const float* topk_weights,
                          float* out_f32) {
    return run_device(h, (hipStream_t)stream, num_tokens, top_k, hidden, topk_ids, topk_weights,
                      out_f32, LKM_DT_F32);
}

extern "C" int lkm_prefill_device(LkmHandle h, const void* hidden, void* out,
                                  const int32_t* topk_ids, const float* topk_weights,
                                  int32_t num_tokens, int32_t top_k, void* stream) {
    LKM_REQUIRE(h, "null engine handle");
    return run_device(h, (hipStream_t)stream, num_tokens, top_k, hidden, topk_ids, topk_weights, out,
                      h->adt);
}

extern "C" int lkm_forward_strided(LkmHandle h, void* stream, int32_t num_tokens, int32_t top_k,
                                   const void* hidden, int64_t hidden_ld, const int32_t* topk_ids,
                                   int64_t ids_ld, int32_t id_offset, const float* topk_weights,
                                   int64_t weights_ld, void* out, int32_t out_dtype) {
    LKM_REQUIRE(h, "null engine handle");
    LKM_REQUIRE(out_dtype == LKM_DT_F32 || out_dtype == h->adt, "forward_strided: out_dtype must be fp32 or the activation dtype");
    LKM_REQUIRE(id_offset >= 0, "forward_strided: id_offset must be >= 0");
    const InLayout il{hidden_ld, ids_ld, weights_ld, id_offset};
    return run_device(h, (hipStream_t)stream, num_tokens, top_k, hidden, topk_ids, topk_weights, out, out_dtype, &il);
}

extern "C" int lkm_forward_routed(LkmHandle h, void* stream, int32_t num_tokens, int32_t top_k, const void* hidden,
                                  int64_t hidden_ld, const void* router_logits, int32_t logits_dtype,
                                  int32_t router_experts, const float* score_bias, int32_t n_group,
                                  int32_t topk_group, int32_t scoring, int32_t renormalize, float routed_scaling,
                                  int32_t id_offset, float* topk_weights_out, int32_t* topk_ids_out, void* out,
                                  int32_t out_dtype) {
    LKM_REQUIRE(h, "null engine handle");
    LKM_REQUIRE(out_dtype == LKM_DT_F32 || out_dtype == h->adt, "forward_routed: out_dtype must be fp32 or the activation dtype");
    LKM_REQUIRE(id_offset >= 0, "forward_routed: id_offset must be >= 0");
    const int M = num_tokens, K = top_k, E = router_experts;
    LKM_REQUIRE(M >= 0 && E > 0 && K > 0 && K <= E && K <= 64, "forward_routed: bad shape M=%d E=%d K=%d", M, E, K);
    if (M == 0) return LKM_OK;
    LKM_REQUIRE(router_logits && topk_weights_out && topk_ids_out, "forward_routed: null device pointer");
    // one chunk, batched path: the router rides in the sort launch; otherwise the two calls it stands for
    // (the chunk size run_device will really use: an engine whose rows pass through aligned scratch rows, hidden_size % 8
    // != 0, chunks at pad_tokens even when the shared arena would hold more -- a fused router on a split batch would
    // route all M rows against one chunk's scratch)
    size_t eff_chunk = chunk_tokens(h, K);
    if (h->Hu != h->H && h->pad_tokens < eff_chunk) eff_chunk = h->pad_tokens;
    bool fused = M > 1 && launch_route_sort_ok(M, K, E, n_group, h->E) && h->t_fuse >= 0 && eff_chunk >= (size_t)M;
    if (M <= direct_max_tokens(h) && h->t_fuse >= 0 && eff_chunk >= (size_t)M) {
        // one to four tokens: the direct path (two launches) routes inside GEMM1 -- when the plan takes that path
        Plan pl;
        pick_cfg(h, M, (size_t)M * K, &pl);
        if (direct_plan(h, M, K, pl)) fused = true;
    }
    InLayout il{hidden_ld, K, K, id_offset};
    if (!fused) {
        int rc = n_group > 0 ? lkm_grouped_topk(stream, router_logits, logits_dtype, score_bias, M, E, K, n_group,
                                                topk_group, scoring, renormalize, routed_scaling, topk_weights_out,
                                                topk_ids_out)
                             : lkm_topk_softmax(stream, router_logits, logits_dtype, score_bias, M, E, K, scoring,
                                                renormalize, routed_scaling, topk_weights_out, topk_ids_out);
        if (rc != LKM_OK) return rc;
        return run_device(h, (hipStream_t)stream, M, K, hidden, topk_ids_out, topk_weights_out, out, out_dtype, &il);
    }
    LKM_REQUIRE(E <= kMaxSlots * 64, "forward_routed: router_experts=%d > %d unsupported", E, kMaxSlots * 64);
    LKM_REQUIRE(logits_dtype >= LKM_DT_F32 && logits_dtype <= LKM_DT_F16, "forward_routed: bad logits dtype");
    LKM_REQUIRE(scoring == 0 || scoring == 1, "forward_routed: scoring must be 0 (softmax) or 1 (sigmoid)");
    if (n_group > 0) {
        LKM_REQUIRE(n_group <= 64 && E % n_group == 0, "forward_routed: n_group=%d must divide E=%d and be <= 64", n_group, E);
        LKM_REQUIRE(topk_group > 0 && topk_group <= n_group, "forward_routed: bad topk_group=%d", topk_group);
        LKM_REQUIRE(K <= topk_group * (E / n_group), "forward_routed: K=%d exceeds kept experts", K);
    }
    RouteArgs ra{};
    ra.src = LogitSrc{router_logits, logits_dtype, 1, 0, nullptr, LKM_DT_F32, nullptr};
    ra.bias = score_bias;
    ra.M = M;
    ra.E = E;
    ra.K = K;
    ra.n_group = n_group > 0 ? n_group : 0;
    ra.topk_group = topk_group;
    ra.scoring = scoring;
    ra.renorm = renormalize;
    ra.rsf = routed_scaling;
    ra.out_w = topk_weights_out;
    ra.out_ids = topk_ids_out;
    il.route = &ra;
    return run_device(h, (hipStream_t)stream, M, K, hidden, topk_ids_out, topk_weights_out, out, out_dtype, &il);
}

extern "C" int lkm_prefill_host(LkmHandle h, int32_t num_tokens, int32_t top_k,
                                const int32_t* topk_ids, const float* topk_weights,
                                const void* hidden, float* out_f32) {
    LKM_REQUIRE(h, "null engine handle");
    LKM_REQUIRE(num_tokens >= 0 && top_k > 0, "bad sizes");
    if (num_tokens == 0) return LKM_OK;
    LKM_REQUIRE(topk_ids && topk_weights && hidden && out_f32, "null host pointer");
    LKM_HIP_CHECK(hipSetDevice(h->device));
    const size_t M = num_tokens;
    if (M > h->io_tokens || !h->io_x) {
        if (h->io_x) (void)hipFree(h->io_x);
        if (h->io_ids) (void)hipFree(h->io_ids);
        if (h->io_w) (void)hipFree(h->io_w);
        if (h->io_out) (void)hipFree(h->io_out);
        h->io_x = h->io_ids = h->io_w = h->io_out = nullptr;
        h->io_tokens = 0;
        LKM_HIP_CHECK(hipMalloc(&h->io_x, M * h->H * 2));      // (H >= Hu: the copies below move Hu-element rows)
        LKM_HIP_CHECK(hipMalloc(&h->io_ids, M * 64 * 4));
        LKM_HIP_CHECK(hipMalloc(&h->io_w, M * 64 * 4));
        LKM_HIP_CHECK(hipMalloc(&h->io_out, M * h->H * 4));
        h->io_tokens = M;
    }
    LKM_REQUIRE(top_k <= 64, "top_k=%d out of range", top_k);
    LKM_HIP_CHECK(hipMemcpy(h->io_x, hidden, M * h->Hu * 2, hipMemcpyHostToDevice));
    LKM_HIP_CHECK(hipMemcpy(h->io_ids, topk_ids, M * top_k * 4, hipMemcpyHostToDevice));
    LKM_HIP_CHECK(hipMemcpy(h->io_w, topk_weights, M * top_k * 4, hipMemcpyHostToDevice));
    int rc = run_device(h, nullptr, num_tokens, top_k, h->io_x, (const int32_t*)h->io_ids,
                        (const float*)h->io_w, h->io_out, LKM_DT_F32);
    if (rc != LKM_OK) return rc;
    LKM_HIP_CHECK(hipMemcpy(out_f32, h->io_out, M * h->Hu * 4, hipMemcpyDeviceToHost));
    return LKM_OK;
}

extern "C" int lkm_per_token_group_quant_fp8(void* stream, const void* x, int32_t x_dtype, int64_t ld_x, int32_t rows,
                                             int32_t cols, void* q, float* scales) {
    LKM_REQUIRE(x && q && scales, "lkm_per_token_group_quant_fp8: null pointer");
    LKM_REQUIRE(x_dtype == LKM_DT_BF16 || x_dtype == LKM_DT_F16, "lkm_per_token_group_quant_fp8: 16-bit activations only");
    LKM_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && ld_x >= cols && ld_x % 8 == 0 && ld_x <= 0x7fffffffLL,
                "lkm_per_token_group_quant_fp8: rows=%d cols=%d ld=%lld (cols and ld multiples of 8)", rows, cols, (long long)ld_x);
    return launch_quant_fp8_rows((hipStream_t)stream, x, (int)ld_x, x_dtype, rows, cols, q, scales);
}

extern "C" int lkm_wna16_expand(void* stream, const void* qweight, const void* scales, const void* zeros, void* out,
                                int64_t rows, int32_t K, int32_t group, int32_t weight_bits, int32_t out_dtype) {
    LKM_REQUIRE(qweight && scales && out, "lkm_wna16_expand: null pointer");
    LKM_REQUIRE(out_dtype == LKM_DT_BF16 || out_dtype == LKM_DT_F16, "lkm_wna16_expand: 16-bit output only");
    LKM_REQUIRE(weight_bits == 4 || weight_bits == 8, "lkm_wna16_expand: weight_bits=%d (4 or 8)", weight_bits);
    LKM_REQUIRE(rows >= 0 && K > 0 && K % 8 == 0 && group > 0 && group % 8 == 0 && K % group == 0,
                "lkm_wna16_expand: rows=%lld K=%d group=%d (K a multiple of the group, the group of 8)", (long long)rows, K, group);
    LKM_REQUIRE(!(zeros && weight_bits == 4 && (rows & 1)), "lkm_wna16_expand: packed 4-bit zero points need an even row count");
    return launch_wna16_expand((hipStream_t)stream, qweight, scales, zeros, out, rows, K, group, weight_bits, out_dtype);
}

extern "C" int lkm_pointer_is_device(const void* p) {
    if (!p) return 0;
    return is_device_ptr(p) ? 1 : 0;
}

extern "C" int lkm_sort_slots(void* stream, const int32_t* ids, int32_t n_slots, int32_t E,
                              int32_t* counts, int32_t* offsets, int32_t* sorted_slot,
                              int32_t* pos_of_slot) {
    LKM_REQUIRE(n_slots >= 0 && E > 0, "sort_slots: bad sizes");
    // active/meta scratch: borrow the tail of a temporary allocation
    int32_t* tmp = nullptr;
    const size_t hist_cap = n_slots > 4096 ? ((size_t)n_slots / 1024 + 1) * E : 0;
    LKM_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(int32_t) * ((size_t)E + kMetaInts + hist_cap)));
    int32_t* hist = hist_cap ? tmp + E + kMetaInts : nullptr;
    int rc = launch_sort((hipStream_t)stream, ids, 1, 1, 0, n_slots, E, counts, offsets, sorted_slot,
                         pos_of_slot, tmp, tmp + E, 0, 0, nullptr, nullptr, hist, hist_cap);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(tmp);
    if (rc != LKM_OK) return rc;
    LKM_HIP_CHECK(e);
    return LKM_OK;
}

extern "C" int lkm_hbm_read_probe(void* stream, const void* device_buf, int64_t bytes, int32_t n_blocks,
                                  int32_t unroll, int32_t reps, float* ms_per_rep) {
    LKM_REQUIRE(device_buf && bytes >= (int64_t)1 << 20 && n_blocks > 0 && reps > 0 && ms_per_rep, "read_probe: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    unsigned* sink = nullptr;
    LKM_HIP_CHECK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    LKM_HIP_CHECK(hipEventCreate(&e0));
    LKM_HIP_CHECK(hipEventCreate(&e1));
    int rc = launch_read_probe(st, device_buf, (size_t)bytes, n_blocks, unroll, sink);   // warm-up
    if (rc == LKM_OK) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < reps && rc == LKM_OK; ++i)
            rc = launch_read_probe(st, device_buf, (size_t)bytes, n_blocks, unroll, sink);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_per_rep = ms / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return rc;
}

extern "C" int lkm_set_profiling(LkmHandle h, int32_t enable) {
    LKM_REQUIRE(h, "null engine handle");
    h->prof = enable != 0;
    h->prof_valid = false;
    return LKM_OK;
}

extern "C" int lkm_get_profile(LkmHandle h, float* ms) {
    LKM_REQUIRE(h && ms, "null argument");
    LKM_REQUIRE(h->prof_valid, "no profiled call recorded (lkm_set_profiling + a decode/prefill call first)");
    LKM_HIP_CHECK(hipEventSynchronize(h->ev[LKM_PROF_N]));
    for (int i = 0; i < LKM_PROF_N; ++i) LKM_HIP_CHECK(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    if (h->prof_rep_used > 1) {   // the GEMM intervals hold prof_rep back-to-back launches
        ms[LKM_PROF_GEMM1] /= (float)h->prof_rep_used;
        ms[LKM_PROF_GEMM2] /= (float)h->prof_rep_used;
    }
    return LKM_OK;
}

extern "C" int64_t lkm_weight_bytes(LkmHandle h) { return h ? h->weight_bytes : 0; }

extern "C" int lkm_describe(LkmHandle h, char* buf, int32_t buf_len) {
    LKM_REQUIRE(h && buf && buf_len > 0, "null argument");
    snprintf(buf, buf_len, "E=%d H=%d I=%d wf=%d%s adt=%d gated=%d T1_half=%d U1=%d T2=%d U2=%d | %s",
             h->E, h->Hu, h->I, h->wf, h->zp ? " zp=1" : "", h->adt, (int)h->gated, h->T1_half, h->U1, h->T2, h->U2,
             h->last_desc);
    return LKM_OK;
}

// "lkm::gemm1_act_kernel<0, 1, 1, 2, true, false, false>" of a host stub: the device function's name as the profilers
// print it (demangled, without return type and parameter list)
#include <cxxabi.h>
static std::string kernel_name_of(const void* host_fn) {
    if (!host_fn) return "";
    const char* m = hipKernelNameRefByPtr(host_fn, nullptr);
    if (!m) {
        (void)hipGetLastError();
        return "?";
    }
    int status = 0;
    char* d = abi::__cxa_demangle(m, nullptr, nullptr, &status);
    std::string n = (status == 0 && d) ? d : m;
    free(d);
    if (n.rfind("void ", 0) == 0) n = n.substr(5);
    // cut the parameter list: the last top-level '(' (template arguments may hold parentheses: "(lkm::Fmt)1")
    int depth = 0;
    size_t cut = std::string::npos;
    for (size_t i = 0; i < n.size(); ++i) {
        if (n[i] == '<') ++depth;
        else if (n[i] == '>') --depth;
        else if (n[i] == '(' && depth == 0) { cut = i; break; }
    }
    if (cut != std::string::npos) n.resize(cut);
    return n;
}

extern "C" int lkm_last_kernels(LkmHandle h, char* buf, int32_t buf_len) {
    LKM_REQUIRE(h && buf && buf_len > 0, "null argument");
    std::string out;
    for (int g = 1; g <= 2; ++g) {
        out += g == 1 ? "gemm1=" : ";gemm2=";
        for (int k = 0; k < 2; ++k)
            if (h->last_fn[g][k]) out += (k ? "+" : "") + kernel_name_of(h->last_fn[g][k]);
    }
    snprintf(buf, buf_len, "%s", out.c_str());
    return LKM_OK;
}

extern "C" int lkm_tuned_plans(LkmHandle h, int64_t* keys, int32_t* index, int32_t cap) {
    if (!h) {
        set_error("lkm_tuned_plans: null engine handle");
        return LKM_E_INVALID;
    }
    int n = 0;
    for (const auto& kv : h->tuned) {         // (std::map: ascending keys -- the same order on every rank)
        if (n < cap && keys && index) {
            keys[n] = (int64_t)kv.first;
            index[n] = kv.second.index;
        }
        ++n;
    }
    return n;
}

extern "C" int lkm_tuned_plan_set(LkmHandle h, int64_t key, int32_t index) {
    LKM_REQUIRE(h, "null engine handle");
    auto it = h->tuned_shape.find((uint64_t)key);
    LKM_REQUIRE(it != h->tuned_shape.end(), "lkm_tuned_plan_set: no plan was tuned for key %lld on this engine", (long long)key);
    const std::vector<LkmEngine::TunedPlan> cands = tune_candidates(h, it->second.M, it->second.K);
    LKM_REQUIRE(index >= 0 && (size_t)index < cands.size(), "lkm_tuned_plan_set: index %d outside the %zu candidates of the shape", index, cands.size());
    LkmEngine::TunedPlan pl = cands[index];
    pl.index = index;
    auto cur = h->tuned.find((uint64_t)key);
    pl.us = (cur != h->tuned.end() && cur->second.index == index) ? cur->second.us : 0.f;     // (0: taken from the group, not timed here)
    h->tuned[(uint64_t)key] = pl;
    return LKM_OK;
}

extern "C" int lkm_set_tuning(LkmHandle h, const char* key, int32_t value) {
    LKM_REQUIRE(h && key, "null argument");
    // every key but the measurement-only ones changes what pick_cfg plans: plans remembered by the autotune are for the
    // old knobs
    if (strcmp(key, "autotune") && strcmp(key, "prof_rep") && strcmp(key, "dbg") && strcmp(key, "valid_den")) {     // ("valid_den" is part of the shape key)
        h->tuned.clear();
        h->tuned_shape.clear();
    }
    if (!strcmp(key, "nt1")) h->t_nt1 = value;
    else if (!strcmp(key, "nt2")) h->t_nt2 = value;
    else if (!strcmp(key, "kw1")) h->t_kw1 = value;
    else if (!strcmp(key, "sk2")) h->t_sk2 = value;
    else if (!strcmp(key, "tbmax")) h->t_tb = value;
    else if (!strcmp(key, "tiled")) h->t_tiled = value;
    else if (!strcmp(key, "tiled2")) h->t_tiled2 = value;
    else if (!strcmp(key, "fuseq")) h->t_fuseq = value;
    else if (!strcmp(key, "waves")) h->t_waves = value;
    else if (!strcmp(key, "pd1")) h->t_pd1 = value;
    else if (!strcmp(key, "pd2")) h->t_pd2 = value;
    else if (!strcmp(key, "xcd")) h->t_xcd = value;
    else if (!strcmp(key, "pf")) h->t_pf = value;
    else if (!strcmp(key, "ydt")) h->t_ydt = value;
    else if (!strcmp(key, "direct")) h->t_direct = value;
    else if (!strcmp(key, "fuse")) h->t_fuse = value;
    else if (!strcmp(key, "autotune")) {
        // Under expert parallelism two ranks timing the same candidates may pick different plans = different fp32
        // summation orders inside one group.  1 is refused there; 2 = "the host agrees on the plans across the group
        // before relying on them" (lvllm_amd.ep.agree_tuned_plans: lkm_tuned_plans -> all_reduce(MIN) -> lkm_tuned_plan_set)
        LKM_REQUIRE(!(value == 1 && h->cfg.num_processes > 1),
                    "lkm_set_tuning: autotune=1 on an expert-parallel engine (num_processes=%d): ranks could choose different "
                    "plans; use 2 and agree on them across the group (lvllm_amd.ep.agree_tuned_plans)", h->cfg.num_processes);
        h->t_autotune = value;
        if (value <= 0) {                        // (0 / -1: off, and forget what was chosen)
            h->tuned.clear();
            h->tuned_shape.clear();
        }
    }
    else if (!strcmp(key, "valid_den")) h->t_valid_den = value;
    else if (!strcmp(key, "hybrid")) h->t_hybrid = value;
    else if (!strcmp(key, "mixed")) h->t_mixed = value;
    else if (!strcmp(key, "prof_rep")) h->t_prof_rep = value;
    else if (!strcmp(key, "dbg")) {
#ifndef LKM_ABLATIONS
        // Bits 1 (4-bit kernels: the packed-fp32-free decoder), 4 (fp8 x fp8 prefill: serialised units), 8 (16-bit prefill:
        // plain run order inside an XCD) and 256 (streamer: per-unit scales) select result-preserving variants, each covered
        // by a parity test.  Every other bit is a TIMING ABLATION that returns wrong numbers by construction: development
        // libraries only (python -m lvllm_amd.build --flag=-DLKM_ABLATIONS), never the liblkm.so that `lk_moe` loads.
        LKM_REQUIRE((value & ~(1 | 4 | 8 | 256)) == 0,
                    "lkm_set_tuning: 'dbg' = %d selects a timing ablation; this library ships none (build with -DLKM_ABLATIONS)", value);
#endif
        h->t_dbg = value;
    }
    else {
        set_error("lkm_set_tuning: unknown key '%s'", key);
        return LKM_E_INVALID;
    }
    return LKM_OK;
}

// ------------------------------------------------------------------ expert images (include/lkm_eplb.h)
// One expert's slabs of the six HBM buffers, in image order: w13, w2, s13, s2, gs13, gs2.  The per-expert
// sizes are the allocation formulas of lkm_create divided by E (every buffer is [E][...] with the expert
// outermost); in the image each slab is padded to 16 bytes.
namespace {
struct ExpertSlabs {
    char* ptr[6];
    size_t bytes[6];
    size_t offset[6];
    size_t total;
};
}  // namespace

static void expert_slabs(const LkmEngine* h, int expert, ExpertSlabs* s) {
    const size_t halves = h->gated ? 2 : 1, loads = wf_loads(h->wf);
    const size_t t13 = halves * h->T1_half * h->U1;   // (16-row tile, k unit) pairs of one expert
    const size_t t2 = (size_t)h->T2 * h->U2;
    size_t sb13 = 0, sb2 = 0;                         // scale bytes of one expert
    if (h->ps) {
        sb13 = t13 * 16 * 4;
        sb2 = t2 * 16 * 4;
    } else if (h->zp) {
        sb13 = t13 * 16 * h->spu * 4;
        sb2 = t2 * 16 * h->spu * 4;
    } else if (h->wf == LKM_W_INT4_B8) {
        sb13 = t13 * 16 * h->spu * 2;
        sb2 = t2 * 16 * h->spu * 2;
    } else if (h->wf == LKM_W_MXFP4 || h->wf == LKM_W_NVFP4) {
        const size_t spu = 128 / h->cfg.groupK;
        sb13 = t13 * 16 * spu;
        sb2 = t2 * 16 * spu;
    } else if (h->wf == LKM_W_FP8_E4M3) {
        sb13 = t13 * 16 * 4;
        sb2 = t2 * 16 * 4;
    }
    void* const base[6] = {h->w13, h->w2, h->s13, h->s2, h->gs13, h->gs2};
    const size_t bytes[6] = {t13 * loads * 64 * 16, t2 * loads * 64 * 16, sb13, sb2, 4, 4};
    size_t off = 0;
    for (int i = 0; i < 6; ++i) {
        s->bytes[i] = base[i] ? bytes[i] : 0;
        s->ptr[i] = base[i] ? (char*)base[i] + (size_t)expert * bytes[i] : nullptr;
        s->offset[i] = off;
        off += (s->bytes[i] + 15) / 16 * 16;
    }
    s->total = off;
}

extern "C" int64_t lkm_expert_bytes(LkmHandle h) {
    if (!h) {
        set_error("lkm_expert_bytes: null engine handle");
        return LKM_E_INVALID;
    }
    ExpertSlabs s;
    expert_slabs(h, 0, &s);
    return (int64_t)s.total;
}

static int copy_expert(LkmHandle h, void* stream, int32_t expert, char* image, bool to_image) {
    LKM_REQUIRE(h, "null engine handle");
    LKM_REQUIRE(image, "null expert image pointer");
    LKM_REQUIRE(expert >= 0 && expert < h->E, "expert %d out of range (engine holds %d local experts)", expert, h->E);
    LKM_HIP_CHECK(hipSetDevice(h->device));
    ExpertSlabs s;
    expert_slabs(h, expert, &s);
    for (int i = 0; i < 6; ++i) {
        if (!s.bytes[i]) continue;
        char* img = image + s.offset[i];
        const size_t pad = (16 - s.bytes[i] % 16) % 16;   // images are deterministic: padding is written as zeros
        if (to_image && pad) LKM_HIP_CHECK(hipMemsetAsync(img + s.bytes[i], 0, pad, (hipStream_t)stream));
        LKM_HIP_CHECK(hipMemcpyAsync(to_image ? (void*)img : (void*)s.ptr[i], to_image ? (const void*)s.ptr[i] : (const void*)img,
                                     s.bytes[i], hipMemcpyDefault, (hipStream_t)stream));      // (the image may be pinned host memory: lvllm_amd/spill.py)
    }
    return LKM_OK;
}

extern "C" int lkm_export_expert(LkmHandle h, void* stream, int32_t expert, void* dst) {
    return copy_expert(h, stream, expert, (char*)dst, true);
}

extern "C" int lkm_import_expert(LkmHandle h, void* stream, int32_t expert, const void* src) {
    return copy_expert(h, stream, expert, (char*)src, false);
}
