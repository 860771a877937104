// Launchers of the fp16 x 2 fused attention (attn_h2.hip), called by the C-ABI entry points in attention.hip.
#pragma once
#include "attn_shared.h"

namespace svl_attn_h2 {

// bytes of the operand workspace: the packed (z = image x head)-major fp16 x 2 operand sets + scale exponents
// (+ the per-query (14 - LSE log2 e, D 2^-g) pairs of the backward)
long ws_bytes(int B, int T, int H, int backward);
// fwd_pack: the pack pass (Q, K row-major; V transposed); fwd: the MFMA grid over `nb` blocks of 256 queries per (image, head)
int fwd(const AttnP& p, int nb, void* ws, long wsb, hipStream_t st);
// pack pass (Q, K, V, dO row-major; Q, K, dO transposed) + D = rowsum(dO * O) (also written to dsum_ws for the leftover-row
// kernels); then the two MFMA grids
int bwd_prepare(const AttnP& p, const float* out, float* dsum_ws, void* ws, long wsb, hipStream_t st);
int bwd_main(const AttnP& p, int nb, void* ws, hipStream_t st);
// the rows past the last full 256-row block (at most 4, from row0) as single-wave MFMA workgroups on the packed operands; they
// read what the pack pass / bwd_prepare wrote: launch them on a stream ordered AFTER those (the helper stream forked behind)
int fwd_pack(const AttnP& p, void* ws, long wsb, hipStream_t st);
int fwd_tail(const AttnP& p, int row0, void* ws, hipStream_t aux);
int bwd_tail(const AttnP& p, int row0, void* ws, hipStream_t aux);

}  // namespace svl_attn_h2
