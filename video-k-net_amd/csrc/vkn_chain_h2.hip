// vkn_chain_h2.hip — the persistent row-owner chain (vkn_chain.hip: k_chain_a / k_chain_c) on the TWO-term fp16 split, hi + lo with the
// three cross products hi.hi + hi.lo + lo.hi: 4 bytes per weight through the L2 -> L1 path and 3 MFMAs per operand pair instead of
// 6 bytes / 6 MFMAs (the bf16 x 3 split).  VERDICT r04 item 4; the split the gather and decode kernels have always used.
//
// Same source, compiled with CH_H2 (every difference hangs on that macro in vkn_chain.hip).  What the fp16 form adds is RANGE MANAGEMENT
// — fp16 spans 2^-24 .. 65504, so a value keeps both halves' eleven bits only between 2^-3 and 2^16:
//   * weight images: every matrix times ONE power of two that puts its maximum at 2^9 .. 2^10 (vkn_pow2_scale_f32 at prepare time;
//     k_split_h2); the inverse travels in the constant block and multiplies the accumulators behind each GEMM — exact;
//   * activation images whose row magnitude the chain does not bound — the raw gather sums / x_feat, the incoming kernels, the
//     attention output, the updator's gate product (input_in x parameters_in) — are scaled ROW BY ROW by the power of two of the
//     row's maximum (a lane of the transposed accumulator tile owns one row: the inverse is one more factor of the same multiply);
//   * LayerNorm outputs, ReLU of those, the FFN's hidden activations are O(1) by construction and travel unscaled.
// Accuracy: the products carry 2^-22 instead of 2^-24; measured against the fp64 oracle the chain's outputs sit where torch's own fp32
// GEMMs sit (profiles/r05_chain_two_term_accuracy.txt: CPU emulation; tests/test_gpu_parity.py: the same bounds as the bf16 form).
#define CH_H2 1
#include "vkn_chain.hip"
