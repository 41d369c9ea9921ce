// criterion_asg.hip -- FullConnectionCriterion and ForceAlignmentCriterion as ONE translation unit, and the ASG criterion's
// one-launch forward pass built from both (round 6).
//
// fl::pkg::speech::ASGLoss = FullConnectionCriterion - ForceAlignmentCriterion (recipes/slimIPL/src/Train.cpp:408-410, :1675).  The
// two criteria's forward scans are independent chains of T / 2 dependent frames; side by side on two streams they cost the caller's
// stream one event record and one event wait per pass, and each of those drains the stream for 5 - 13 us
// (profiles/r06_run56_asg_timelines.txt) -- 11 us of a 223 us forward pass.  Two kernels of one stream cannot overlap on this part
// (every launch carries the barrier bit; hipExtAnyOrderLaunch is not honoured on gfx9: tools/micro/any_order_launch.hip).  So the four
// half scans of a forward pass are ONE launch here: workgroup (b, y) runs, for y = 0, 1, the alpha / beta half of
// FullConnectionCriterion (fcc_mitm_fwd_body, threads 0 .. 127) and, for y = 2, 3, the alpha / beta half of ForceAlignmentCriterion
// (fac_half_fwd) -- the same code the two criteria launch by themselves, hence bit-identical results -- and one finish launch behind
// it computes both losses, recomputes what the range checks flagged, and subtracts.  No side stream, no events, three launches:
//   fac_rows_k (+ target sizes, + partials fill)  ->  asg_mitm_fwd  ->  asg_finish_fwd.
// The sources of the two criteria are included here (they are not compiled on their own: csrc/Makefile) because a kernel is one
// function and their kernels' bodies live in those files.
#include "criterion_fcc.hip"
#include "criterion_fac.hip"

namespace w2l {

template <int NW>
__global__ __launch_bounds__(64 * NW > 128 ? 64 * NW : 128) void asg_mitm_fwd(int T, int N, int L, const float* __restrict__ x,
                                                                               const int* __restrict__ target, const int* __restrict__ targetSize,
                                                                               const float* __restrict__ trans, FccWs fcc, FacWs fac) {
  if (blockIdx.y < 2) {   // FullConnectionCriterion: alpha over frames 0 .. m (y = 0), beta over T - 1 .. m (y = 1); two waves
    if (threadIdx.x >= 128) return;   // (a wave that has ended does not take part in the workgroup's barriers)
    fcc_mitm_fwd_body(T, N, x, trans, fcc, (int)blockIdx.y);
    return;
  }
  __shared__ FacRec ring[NW][kPlinRing];
  __shared__ int prog[NW];
  if (threadIdx.x >= 64 * NW) return;   // (NW = 1: the workgroup has two waves for the other criterion's role)
  if (targetSize[blockIdx.x] <= 0) return;   // the finish launch writes loss 0
  if (threadIdx.x < NW) prog[threadIdx.x] = -1;
  // workgroup barrier among the NW waves of this role (the others have ended)
  __syncthreads();
  if (blockIdx.y == 2) fac_half_fwd<NW, false>(T, N, L, target, targetSize, trans, fac, ring, prog, 0);
  else fac_half_fwd<NW, true>(T, N, L, target, targetSize, trans, fac, ring, prog, 0);
}

// both criteria's finish in one launch: wave 0 first computes FullConnectionCriterion's loss from its two halves (or the whole
// log-domain recursion of an utterance its range check flagged: fcc_fwd_log), then the workgroup runs fac_mitm_finish_all's sequence
// (ForceAlignmentCriterion's loss, its flagged-utterance recomputation) and thread 0 subtracts
__global__ __launch_bounds__(kFacFinishThreads) void asg_finish_fwd(int T, int N, int L, int scaleMode, const float* __restrict__ x,
                                                                    const int* __restrict__ target, const int* __restrict__ targetSize,
                                                                    const float* __restrict__ trans, float* loss, float* loss2, FccWs fcc, FacWs fac) {
  const int b = blockIdx.x;
  if (threadIdx.x < 64) fcc_fwd_log_body<32, true>(T, N, scaleMode, x, targetSize, trans, loss, fcc);
  fac_mitm_finish_body(T, N, L, scaleMode, target, targetSize, trans, loss2, fac);
  __syncthreads();   // the flag and the losses of wave 0, visible to the workgroup
  if (*(volatile int*)(fac.redo + b) != 0) {   // workgroup-uniform
    fac_fwd_blk_body<8, 1>(T, N, L, scaleMode, x, target, targetSize, trans, loss2, fac, nullptr);
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[b] = *(volatile float*)(loss + b) - *(volatile float*)(loss2 + b);
}

bool asg_forward_merged_ok(int B, int T, int N, int L) {
  return fac_asg_fused_ok(B, T, N, L) && asg_dpp_path(N) && asg_mitm_path() && mitm_only() < 0 && 4 * B <= 256;
}

// ts [B] (out), loss [B] (out), loss2 [B] (ForceAlignmentCriterion's own loss, kept for diagnostics); the two criteria's workspaces
int asg_forward_merged(int B, int T, int N, int L, int scaleMode, const float* input, const int* target, int* ts, const float* trans,
                       float* loss, float* loss2, void* fccWorkspace, void* facWorkspace, hipStream_t s) {
  if (!asg_forward_merged_ok(B, T, N, L)) return W2L_EUNSUPPORTED;
  if (!input || !target || !ts || !trans || !loss || !loss2 || !fccWorkspace || !facWorkspace) return W2L_EINVAL;
  FccWs fw = fcc_ws(fccWorkspace, B, T, N);
  FacWs aw = fac_ws(facWorkspace, B, T, N, L);
  hipLaunchKernelGGL(fac_rows_k, dim3((unsigned)((T + kFacRowsPerWave * kFacRowsWaves - 1) / (kFacRowsPerWave * kFacRowsWaves)), (unsigned)B), dim3(64 * kFacRowsWaves), 0, s, T, N, input,
                     trans, aw.crow, aw.zmax, aw.zspr, target, L, ts, aw.tgpart, (unsigned)((size_t)2 * B * N * N));
  W2L_LAUNCH_CHECK();
#define W2L_ASG_M_GO(NWV) hipLaunchKernelGGL((asg_mitm_fwd<NWV>), dim3(B, 4), dim3(64 * NWV > 128 ? 64 * NWV : 128), mitm_excl(B, (const void*)asg_mitm_fwd<NWV>), s, T, N, L, input, target, (const int*)ts, trans, fw, aw)
  switch ((L + 63) / 64) {
    case 1: W2L_ASG_M_GO(1); break;
    case 2: W2L_ASG_M_GO(2); break;
    case 3: W2L_ASG_M_GO(3); break;
    case 4: W2L_ASG_M_GO(4); break;
    default: W2L_ASG_M_GO(5); break;
  }
#undef W2L_ASG_M_GO
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(asg_finish_fwd, dim3(B), dim3(kFacFinishThreads), 0, s, T, N, L, scaleMode, input, target, (const int*)ts, trans, loss, loss2, fw, aw);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
