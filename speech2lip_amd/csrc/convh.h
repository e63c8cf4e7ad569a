// Arguments of s2l::convh_asm_kernel (csrc/convh.hip): the generated-assembly 3x3 convolution on half-width (bf16) tensors.  The
// assembly body loads the fields from the kernarg segment by offset (gen_convh_body.py: ARG) -- keep the order.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace s2l {

struct ConvHArgs {
  const uint16_t* inA;     // [F][CA/32][H][W][32] bf16: 32-channel planes ("C32", csrc/unet_half.inc)
  const uint16_t* inB;     // [F][CB/32][H][W][32] or null (virtual concat: planes of A first)
  const uint16_t* w16;     // s2l_unet_pack16 chunks of this layer: [rows/64][k/32][tap 9][k-step 2][block 2][lane 64][8]
  const float* bias;       // [cout] or null
  uint16_t* out;           // [F][cout/32][H][W][32] bf16
  const uint16_t* gate;    // or null: out's shape; out = gate > 0 ? value : 0
  int CA, CB, cout, H, W, tiles_x, tiles_y, n_ct;      // tiles of 32 rows x 16 columns
  int relu;                // 1: max(0, .) before the conversion
  int n_frames;
  float* stat;             // or null: per-tile partial sums of BatchNorm's batch statistics of the STORED values, channel_stats_h_kernel's
                           //   layout with block = tile: stat[((frame * tiles_xy + ty * tiles_x + tx) * 2 + {sum, sum of squares}) * cout + channel]
  const float* norm;       // or null: inA is a PRE-BatchNorm tensor z and the convolution consumes a = relu(z * scale + shift): per frame 512 floats,
                           //   scale at [c], shift at [CA + c] (bn_finalize_groups_kernel's row).  The halo tile is normalised in LDS with
                           //   bn_relu_h_kernel's expression and rounding, so the output is the bits of the two-kernel route.  CB = 0, no gate, no ReLU.
  const uint16_t* bz;      // or null (with bst and stat: BACKWARD statistics): the pre-BatchNorm tensor z of the layer whose OUTPUT gradient this
                           //   (input-gradient) convolution produces, out's shape.  The launch then leaves, per tile and channel, sum g' and sum g' z
                           //   with g' = fma(z, scale, shift) > 0 ? g : 0 on the stored g (stage 1 of that layer's BatchNorm backward,
                           //   bn_bwd_reduce_h_kernel's expressions) in stat's layout; out itself is stored unmasked.  No gate, no ReLU, no norm.
  const float* bst;        //   that layer's per-frame rows (512 floats: scale at [c], shift at [cout + c])
};
constexpr int kConvHStatBlocks = 1024;      // tiles per frame the statistics buffers hold (= kStatBlocks of csrc/unet.hip)

// 0 if the launch was taken (*launched) or does not fit the kernel (!*launched).  a.stat != null: *stats_done says whether the kernel that ran
// left the statistics (the default eight-wave form, no gate, no ReLU, at most kConvHStatBlocks tiles per frame); otherwise the caller runs
// its own pass.  *stat_blocks: tiles per frame.
int launch_convh(const ConvHArgs& a, hipStream_t st, bool* launched, bool* stats_done = nullptr, int* stat_blocks = nullptr);

}  // namespace s2l
