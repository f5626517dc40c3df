// Epilogue helpers shared by the fp32 (k_conv_fp32.hip) and the split-form (k_conv.hip) convolution kernels: GroupNorm statistics and
// GroupNorm-backward sums of a wave's 32-pixel x 32-channel output tile, computed while the tile is still in registers.
#pragma once
#include "pidm_common.h"

namespace pidm {
// GroupNorm partial statistics of a wave's 32-pixel x 32-channel output tile (ConvGeom::gn_part): each lane holds the sums over
// its 16 rows of one channel; butterfly over the gn_cpg lanes of a group and the two lane halves, one lane per group writes.
#define PIDM_GN_PARTIAL(s1_, s2_, b_, pix_in_img_, c_)                                                              \
  {                                                                                                                 \
    float a1__ = (s1_), a2__ = (s2_);                                                                               \
    for (int off__ = 1; off__ < g.gn_cpg; off__ <<= 1) {                                                            \
      a1__ += __shfl_xor(a1__, off__);                                                                              \
      a2__ += __shfl_xor(a2__, off__);                                                                              \
    }                                                                                                               \
    a1__ += pidm_other_half(a1__);                                                                                   \
    a2__ += pidm_other_half(a2__);                                                                                   \
    if (half == 0 && (l31 & (g.gn_cpg - 1)) == 0) {                                                                 \
      double* o__ = g.gn_part + (((size_t)(b_) * g.gn_nchunk + ((pix_in_img_) >> 5)) * g.gn_G + (c_) / g.gn_cpg) * 2; \
      o__[0] = (double)a1__;                                                                                        \
      o__[1] = (double)a2__;                                                                                        \
    }                                                                                                               \
  }
// GroupNorm-backward partial sums from a dgrad epilogue (ConvGeom::bn_part).  acc_ = the lane's 16 rows of dy for channel c_
// (rows (r&3) + 8(r>>2) + 4half of the wave's 32 pixels), xrow_ = &x[first pixel of the wave][c_], xstep_ = floats between
// consecutive pixels of the wave in x.  Same arithmetic as gn_recompute (k_norm.hip).
// res_on_ (wave-uniform, from kernel arguments): the epilogue adds a residual; rrow_ is evaluated only then.  The optional loads
// (FiLM rows, residual rows) sit behind SCALAR conditions as batches: with a per-lane `ptr ? ptr[..] : 0` hipcc gave each of the 16
// residual rows a branch, a reload of its spilled offset and s_waitcnt vmcnt(0) - 16 memory round trips in the epilogue of every
// input-gradient launch that leaves these sums (ISA of conv3x3_split_ws_kernel, round 6).
#define PIDM_BN_PARTIAL(acc_, bv_, b_, pix_in_img_, c_, xrow_, xstep_, res_on_, rrow_, rstep_)                      \
  {                                                                                                                 \
    const int gI__ = (c_) / g.bn_cpg;                                                                               \
    const float mean__ = g.bn_stats[((size_t)(b_) * g.bn_G + gI__) * 2], rstd__ = g.bn_stats[((size_t)(b_) * g.bn_G + gI__) * 2 + 1]; \
    const float gm__ = g.bn_gamma[(c_)], bt__ = g.bn_beta[(c_)];                                                    \
    float sc__ = 1.f, sh__ = 0.f;                                                                                   \
    if (g.bn_ss) {                                                                                                  \
      sc__ = 1.f + (g.bn_ss[(size_t)(b_) * g.bn_ldss + (c_)] + g.bn_ssb[(c_)]);                                     \
      sh__ = g.bn_ss[(size_t)(b_) * g.bn_ldss + g.Cout + (c_)] + g.bn_ssb[g.Cout + (c_)];                           \
    }                                                                                                               \
    float xv__[16], rv__[16];                                                                                       \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                  \
        xv__[r] = (xrow_)[(unsigned)((r & 3) + 8 * (r >> 2) + 4 * half) * (unsigned)(xstep_)];   /* 32-bit row offsets */ \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) rv__[r] = 0.f;                                                   \
    if (res_on_) {                                                                                                  \
      const float* rr__ = (rrow_);                                                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                \
          rv__[r] = rr__[(unsigned)((r & 3) + 8 * (r >> 2) + 4 * half) * (unsigned)(rstep_)];                       \
    }                                                                                                               \
    float a1__ = 0.f, a2__ = 0.f;                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                \
      const float xh__ = (xv__[r] - mean__) * rstd__;                                                               \
      const float v__ = (xh__ * gm__ + bt__) * sc__ + sh__;                                                         \
      const float sg__ = pidm_sigmoid(v__);                                                                         \
      const float dv__ = (((acc_)[r] + (bv_)) + rv__[r]) * (sg__ * (1.f + v__ * (1.f - sg__)));                     \
      a1__ += dv__;                                                                                                 \
      a2__ += dv__ * xh__;                                                                                          \
    }                                                                                                               \
    a1__ += pidm_other_half(a1__);                                                                                   \
    a2__ += pidm_other_half(a2__);                                                                                   \
    if (half == 0) {                                                                                                \
      double* o__ = g.bn_part + (((size_t)(b_) * g.bn_nchunk + ((pix_in_img_) >> 5)) * g.Cout + (c_)) * 2;          \
      o__[0] = (double)a1__;                                                                                        \
      o__[1] = (double)a2__;                                                                                        \
    }                                                                                                               \
  }
}  // namespace pidm
