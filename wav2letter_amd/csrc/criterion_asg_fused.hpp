// criterion_asg_fused.hpp -- the ASG criterion's fused launch sequence for small label sets (N <= 32, L <= 320: the letter recipes),
// internal to libw2l_hip.so: criterion_host.cpp (AsgSequence: fl::pkg::speech::ASGLoss = FullConnectionCriterion -
// ForceAlignmentCriterion, recipes/slimIPL/src/Train.cpp:408-410, :1675; also behind w2l_asg_forward / w2l_asg_backward) drives it,
// criterion_fac.hip implements it.  The C-ABI entry points of the single criteria (w2l_fac_forward / w2l_fac_backward / w2l_fcc_*,
// include/w2l_hip.h) keep their one-criterion-one-call meaning; what this sequence removes are the launches that exist only because
// ASG composes two such calls:
//   forward   batch_target_size (rides on the label-row pre-pass), the flagged-utterance launch (folded into the finish launch),
//             the loss axpy (the finish launch subtracts from FullConnectionCriterion's loss);
//   backward  the sort of the positions by label (inside the backward scan launch, in front of its shorter half), the clearing of
//             the transition-gradient partials (by forward's label-row launch); the partials' reduce and the two axpy launches are
//             ONE launch behind the join.
//             (Measured and dropped: the scatter subtracting in place from FullConnectionCriterion's input gradient -- at N = 30
//             that criterion's backward scan is the LONGER one, 170 against 152 us, and a wait on another stream's event costs the
//             waiting stream ~7 - 13 us: profiles/r06_run56_asg_timelines.txt.)
// Results are bit-identical to the composed calls (the same operations on the same operands in the same order).  At B = 64,
// T = 2000, N = 30, L <= 300: forward 0.239 -> 0.2245 ms, forward + backward 0.479 -> 0.460 ms (profiles/r06_run59_*); with the
// one-launch forward pass at the end of this file (criterion_asg.hip) 0.211 / 0.449 ms (profiles/r06_run65_*).
#pragma once
#include <hip/hip_runtime.h>

namespace w2l {

// hook(arg, what): called by the sequences below where the caller's second stream has to be forked from / joined to `s`
enum AsgHookPoint {
  ASG_TARGET_SIZES_QUEUED = 0,   // forward: the launch that writes ts is on `s` -- fork FullConnectionCriterion's stream here
  ASG_NEED_FCC_LOSS = 1,         // forward: the next launch subtracts from FullConnectionCriterion's loss
  ASG_NEED_FCC_GRADS = 2,        // backward: the next launch (on sOut) combines the two criteria's gradients: both streams joined
};
typedef void (*AsgHook)(void* arg, int what);

// true where the fused sequence exists (the meet-in-the-middle scans: N <= 32 labels, L <= 320, transition-gradient partials)
bool fac_asg_fused_ok(int B, int T, int N, int L);
// forward: ts[b] (target sizes) and minuend[b] -= ForceAlignmentCriterion loss; `loss2` receives that loss itself
int fac_forward_asg(int B, int T, int N, int L, int scaleMode, const float* input, const int* target, int* ts, const float* trans,
                    float* loss2, float* minuend, void* workspace, hipStream_t s, AsgHook hook, void* arg);
// backward: dEm -= input gradient (through the scratch dx2 [B][T][N]), dTrans -= transition gradient, in one launch behind the
// hook; partialsClear: no backward pass has run on this workspace since fac_forward_asg cleared the transition-gradient partials;
// fccPart (may be null): FullConnectionCriterion's per-utterance partials [b * fccStride][N][N] (fcc_backward_impl, partialsOnly) --
// the launch then also sums those over the utterances instead of reading that criterion's finished transition gradient from dTrans
int fac_backward_asg(int B, int T, int N, int L, const int* target, const int* ts, const float* grad, float* dEm, float* dTrans,
                     float* dx2, void* workspace, bool partialsClear, const float* fccPart, int fccStride, hipStream_t s, hipStream_t sOut,
                     AsgHook hook, void* arg);   // scan + scatter on `s`, the combine launch (behind the hook) on `sOut`
// w2l_fcc_backward; partialsOnly: without its last launch (the sum of the transition-gradient partials over the utterances)
int fcc_backward_impl(int B, int T, int N, const float* trans, const float* grad, float* inputGrad, float* transGrad, void* workspace,
                      hipStream_t stream, bool partialsOnly);
const float* fcc_transgrad_partials(void* workspace, int B, int T, int N, int* stride);   // null: N > 64 (no such partials)

// One-launch forward pass (criterion_asg.hip): label rows (+ target sizes, + partials fill) -> the four half scans of the two criteria
// in ONE launch -> both finishes and the difference in one launch; no side stream.  ok: the fused sequence's conditions and
// FullConnectionCriterion on its meet-in-the-middle scans (N <= 31) with a CU per scan workgroup (4 B <= 256).
bool asg_forward_merged_ok(int B, int T, int N, int L);
int asg_forward_merged(int B, int T, int N, int L, int scaleMode, const float* input, const int* target, int* ts, const float* trans,
                       float* loss, float* loss2, void* fccWorkspace, void* facWorkspace, hipStream_t s);

}  // namespace w2l
