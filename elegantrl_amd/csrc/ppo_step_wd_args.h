// K6 for net_dims = (256, h2): kernel arguments and the per-shape launchers (ppo_step_wd_{14,24,12,22}.hip: KX input tiles of 32 x
// N2 = h2 / 32, one translation unit each so that the shapes compile in parallel)
#pragma once
#include "ppo_step.h"

struct PpoWdArgs {
    Ppo2Args g;          // w2img: the four quarter images, contiguous; w1img: [256][3][K1]
    float *scratch;      // [n_slabs][2 networks] blocks of wd_scratch_floats(): GELU'(z1) tiles | last hidden layer's tiles | four H1 quarter images [| H2, dZ3 images]
    const unsigned char *w3img[2];   // three hidden layers (h3 > 0): the third layer's weight image [h3][3][128 bf16]; else unused
    int h3;
};
int erl_ppo_wd_launch_14(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_wd_launch_24(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_wd_launch_12(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_wd_launch_22(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
// net_dims (256, 128, 64 | 128): ppo_step_wd3_{12,14,22,24}.hip (KX, N3 = h3 / 32)
int erl_ppo_wd3_launch_12(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_wd3_launch_14(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_wd3_launch_22(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);
int erl_ppo_wd3_launch_24(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream);

// per workgroup and network: (8 + N2) register tiles [16][256 threads] fp32, then H1 as four sample-major quarter images
// [128 samples][3 parts][64 bf16] (the LDS layout of the dW2 operand: they come back by LDS-DMA)
constexpr int kWdH1ImgBytes = 128 * 384;
// (NL = tiles of the hidden layer that feeds the output layer; N3 > 0: three hidden layers -- + the H2 image [128][3][128] and the dZ3 image
// [128][3][32 N3], both in lane order)
__host__ __device__ constexpr size_t wd_scratch_floats(int NL, int N3)
{
    return (size_t)(8 + NL) * 16 * 256 + 4 * kWdH1ImgBytes / 4 + (N3 > 0 ? (128 * 768 + 128 * 48 * 4 * N3) / 4 : 0);
}
