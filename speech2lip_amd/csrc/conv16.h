// Arguments of s2l::conv16_asm_kernel (csrc/conv16.hip): the generated-assembly split-bf16 3x3 convolution.  The assembly body
// loads the fields from the kernarg segment by offset (gen_conv16_body.py: ARG) -- keep the order.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace s2l {

struct Conv16Args {
  const float* inA;        // [F,H,W,CA] fp32
  const float* inB;        // [F,H,W,CB] or null (virtual concat: channels of A first)
  const uint16_t* w16;     // s2l_unet_pack16x3 chunks of this layer: [cout/64][cin/16][tap 9][part 2][block 2][lane 64][8]
  const float* bias;       // [cout]
  float* out;              // [F,H,W,cout]
  int CA, CB, cout, H, W, tiles_x, tiles_y, n_ct;      // tiles of 16 x 16 pixels
  int relu;                // 1: ReLU in the epilogue
  int n_frames;
  float* pool;             // or null: MaxPool2d(2) of `out`, [F,H/2,W/2,cout] (a second kernel: the max of the same fp32 values)
};

int launch_conv16_asm(const Conv16Args& a, hipStream_t st, bool* launched);

}  // namespace s2l
