// oracle/_ref (test infrastructure only): exposes the reference's OWN in-tree CPU fused-MoE kernel
// (/root/reference/csrc/cpu/cpu_fused_moe.cpp:640-702, compiled from where it lies) as torch ops in the
// namespace `lkm_ref`, so that tests and bench.py's cpu_baseline can run it next to the oracle and the HIP
// path.  The reference registers the same two functions as torch.ops._C.* in csrc/cpu/torch_bindings.cpp:617-627;
// that file drags in the whole CPU backend (oneDNN, attention, ...), so the two declarations are restated here.
#include <optional>
#include <string>

#include <torch/library.h>
#include <ATen/ATen.h>

void prepack_moe_weight(const at::Tensor& weight, at::Tensor& packed_weight, const std::string& isa);
void cpu_fused_moe(at::Tensor& output, const at::Tensor& input, const at::Tensor& w13, const at::Tensor& w2,
                   const std::optional<at::Tensor>& w13_bias, const std::optional<at::Tensor>& w2_bias,
                   const at::Tensor& topk_weights, const at::Tensor& topk_id, const bool skip_weighted,
                   const std::string& act, const std::string& isa);

TORCH_LIBRARY(lkm_ref, m) {
  m.def("prepack_moe_weight(Tensor weight, Tensor(a1!) packed_weight, str isa) -> ()");
  m.def("cpu_fused_moe(Tensor(a0!) output, Tensor input, Tensor w13, Tensor w2, Tensor? w13_bias, "
        "Tensor? w2_bias, Tensor topk_weights, Tensor topk_id, bool skip_weighted, str act, str isa) -> ()");
}
TORCH_LIBRARY_IMPL(lkm_ref, CPU, m) {
  m.impl("prepack_moe_weight", &prepack_moe_weight);
  m.impl("cpu_fused_moe", &cpu_fused_moe);
}
