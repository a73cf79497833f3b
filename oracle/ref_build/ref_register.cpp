// oracle/_ref (test infrastructure only): exposes the reference's OWN in-tree CPU fused-MoE kernel
// (/root/reference/csrc/cpu/cpu_fused_moe.cpp:640-702, compiled from where it lies) as torch ops in the
// namespace `lkm_ref`, so that tests and bench.py's cpu_baseline can run it next to the oracle and the HIP
// path.  The reference registers the same two functions as torch.ops._C.* in csrc/cpu/torch_bindings.cpp:617-627;
// that file drags in the whole CPU backend (oneDNN, attention, ...), so the two declarations are restated here.
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include <torch/library.h>
#include <ATen/ATen.h>

void prepack_moe_weight(const at::Tensor& weight, at::Tensor& packed_weight, const std::string& isa);
void cpu_fused_moe(at::Tensor& output, const at::Tensor& input, const at::Tensor& w13, const at::Tensor& w2,
                   const std::optional<at::Tensor>& w13_bias, const std::optional<at::Tensor>& w2_bias,
                   const at::Tensor& topk_weights, const at::Tensor& topk_id, const bool skip_weighted,
                   const std::string& act, const std::string& isa);

// Quantised experts: the reference's second in-tree CPU MoE kernel (csrc/cpu/sgl-kernels/moe.cpp:874-1224 with
// moe_fp8.cpp / gemm.cpp:616-727; registered by the reference at csrc/cpu/torch_bindings.cpp:476-492), used here
// for FP8_W8A16 (128x128 block scales) and MXFP4 experts.
at::Tensor convert_weight_packed(at::Tensor& weight);
at::Tensor convert_scale_packed(at::Tensor& scale);
// int4 (GPTQ / AWQ int32-packed) experts -> the kernel's blocked layout (csrc/cpu/sgl-kernels/gemm_int4.cpp:854,
// registered at csrc/cpu/torch_bindings.cpp:531-535); the kernel then computes W4A8 (INT4_W4A8)
std::tuple<at::Tensor, at::Tensor, at::Tensor> convert_weight_packed_scale_zp(at::Tensor qweight, at::Tensor qzeros,
                                                                              at::Tensor scales, int64_t quant_method_4bit);
at::Tensor fused_experts_cpu(at::Tensor& hidden_states, at::Tensor& w1, at::Tensor& w2, at::Tensor& topk_weights,
                             at::Tensor& topk_ids, bool inplace, int64_t moe_comp_method,
                             const std::optional<at::Tensor>& w1_scale, const std::optional<at::Tensor>& w2_scale,
                             const std::optional<at::Tensor>& w1_zero, const std::optional<at::Tensor>& w2_zero,
                             const std::optional<std::vector<int64_t>> block_size,
                             const std::optional<at::Tensor>& w1_bias, const std::optional<at::Tensor>& w2_bias,
                             const std::optional<double>& alpha, const std::optional<double>& limit, bool is_vnni);

TORCH_LIBRARY(lkm_ref, m) {
  m.def("convert_weight_packed(Tensor weight) -> Tensor");
  m.def("convert_scale_packed(Tensor scale) -> Tensor");
  m.def("convert_weight_packed_scale_zp(Tensor weight, Tensor qzeros, Tensor scales, int quant_method_4bit) -> "
        "(Tensor, Tensor, Tensor)");
  m.def("fused_experts_cpu(Tensor hidden_states, Tensor w1, Tensor w2, Tensor topk_weights, Tensor topk_ids, "
        "bool inplace, int moe_comp_method, Tensor? w1_scale, Tensor? w2_scale, Tensor? w1_zero, Tensor? w2_zero, "
        "int[]? block_size, Tensor? w1_bias, Tensor? w2_bias, float? alpha, float? limit, bool is_vnni) -> Tensor");
  m.def("prepack_moe_weight(Tensor weight, Tensor(a1!) packed_weight, str isa) -> ()");
  m.def("cpu_fused_moe(Tensor(a0!) output, Tensor input, Tensor w13, Tensor w2, Tensor? w13_bias, "
        "Tensor? w2_bias, Tensor topk_weights, Tensor topk_id, bool skip_weighted, str act, str isa) -> ()");
}
TORCH_LIBRARY_IMPL(lkm_ref, CPU, m) {
  m.impl("prepack_moe_weight", &prepack_moe_weight);
  m.impl("cpu_fused_moe", &cpu_fused_moe);
  m.impl("convert_weight_packed", &convert_weight_packed);
  m.impl("convert_scale_packed", &convert_scale_packed);
  m.impl("convert_weight_packed_scale_zp", &convert_weight_packed_scale_zp);
  m.impl("fused_experts_cpu", &fused_experts_cpu);
}
