// feeder.hip -- GPU-resident CIFAR-10 batch assembly: gather `B` images by index from the
// uint8 dataset held in HBM, convert to fp32 / 255 (torchvision ToTensor) and, for training, apply the reference's
// augmentation RandomCrop(32, padding=4) + RandomHorizontalFlip (/root/reference/dataloaders.py:23-29) with a
// counter-based per-sample RNG, so a step's batch never touches the host.  HBM-bound byte work:
// 3 KiB read + 12 KiB written per image.
#include "gmk_common.h"

__host__ __device__ __forceinline__ unsigned gmk_mix32(unsigned x) {  // lowbias32 avalanche hash
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// one thread = 4 consecutive output pixels of one (image, channel, row)
__global__ __launch_bounds__(256) void cifar_batch_kernel(const unsigned char* __restrict__ data, const long long* __restrict__ index,
                                                          float* __restrict__ out, long long B, int C, int H, int W,
                                                          int pad, int augment, unsigned seed_lo, unsigned seed_hi) {
  const int W4 = W >> 2;
  const long long total = B * C * H * W4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int x4 = (int)(e % W4);
    long long r = e / W4;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const long long b = r / C;
    const long long src = index[b];
    int dy = pad, dx = pad, flip = 0;
    if (augment) {
      // per-sample draw keyed by (seed, position in the batch stream): dy, dx uniform in [0, 2 pad], flip a fair bit
      const unsigned hsh = gmk_mix32(gmk_mix32((unsigned)b * 0x9E3779B1u + seed_lo) + seed_hi * 0x85EBCA6Bu);
      dy = (int)(hsh % (unsigned)(2 * pad + 1));
      dx = (int)((hsh >> 8) % (unsigned)(2 * pad + 1));
      flip = (int)((hsh >> 16) & 1u);
    }
    const int sy = y + dy - pad;
    const unsigned char* plane = data + ((src * C + c) * H) * (long long)W;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = x4 * 4 + k;
      const int cx = flip ? (W - 1 - x) : x;  // flip acts on the cropped image
      const int sx = cx + dx - pad;
      const bool ok = (sy >= 0) & (sy < H) & (sx >= 0) & (sx < W);
      const unsigned char u = ok ? plane[sy * W + sx] : (unsigned char)0;  // zero padding of RandomCrop
      v[k] = (float)u / 255.0f;                                            // ToTensor: exact fp32 division
    }
    *(float4*)(out + (((b * C + c) * H + y) * (long long)W) + x4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int gmk_cifar_batch(const unsigned char* data, const int64_t* index, float* out, int64_t B, int C, int H,
                               int W, int pad, int augment, uint64_t seed, gmk_stream_t stream) {
  GMK_REQUIRE(data && index && out, GMK_E_NULL);
  GMK_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (W % 4) == 0 && pad >= 0 && pad < H && pad < W, GMK_E_SHAPE);
  GMK_REQUIRE(gmk_aligned16(out), GMK_E_ALIGN);
  const long long total = (long long)B * C * H * (W / 4);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cifar_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, data,
                     (const long long*)index, out, (long long)B, C, H, W, pad, augment, (unsigned)(seed & 0xFFFFFFFFu),
                     (unsigned)(seed >> 32));
  GMK_CHECK_LAUNCH();
  return GMK_OK;
}
