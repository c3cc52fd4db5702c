/* gmk.h -- C ABI of libgmk.so: hand-written gfx950 (MI355X) kernels for the hot path of
 * michaelklachko/generative-models: the CIFAR-10 conv autoencoder / VAE train step and the
 * FID / inception feature statistics.
 *
 * The reference has no FFI of its own (it is eager PyTorch); each entry point below replaces
 * the PyTorch call(s) cited in its comment (file:line under /root/reference).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; every buffer (inputs, outputs, workspace) is DEVICE memory owned by
 *     the caller; the library never allocates, frees or retains a pointer;
 *   - tensors are fp32, contiguous, NCHW (activations), OIHW (conv weights), IOHW (transposed-conv
 *     weights), [out][in] (linear weights): the layouts of the reference's state_dict;
 *   - asynchronous launch on `stream` (a hipStream_t passed as void*; NULL = default stream), no internal
 *     synchronisation, re-entrant across streams and devices.  State the library keeps, all of it listed here: (1) the
 *     per-device kernel-path selection of gmk_conv_set_thin_paths (one atomic word per device); (2) per-device caches that
 *     only ever grow (event ring of gmk_stream_wait_stream, raised dynamic-LDS limits, the address of the static scratch of
 *     gmk_vae_loss_fwd_bwd_static); (3) the calling thread's plan being recorded between gmk_plan_begin / gmk_plan_end;
 *     (4) profiling knobs read once from the environment (GMK_WINO, GMK_EVENT_SCOPE, GMK_*_DBG: ablation builds, 0 in production);
 *   - return 0 on success, a negative GMK_E_* on a rejected argument, a positive hipError_t if a
 *     launch failed; gmk_strerror() names either.
 *   - spatial sizes must be powers of two (the reference fixes 32x32; 64x64 for config 5).
 */
#ifndef GMK_H
#define GMK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gmk_stream_t;

enum gmk_error {
  GMK_OK = 0,
  GMK_E_SHAPE = -1,       /* unsupported / inconsistent shape */
  GMK_E_ALIGN = -2,       /* pointer not 16-byte aligned */
  GMK_E_UNSUPPORTED = -3, /* unsupported option (kernel size, activation, mode) */
  GMK_E_WORKSPACE = -4,   /* workspace too small */
  GMK_E_NULL = -5,        /* required pointer is NULL */
  GMK_E_TOO_LARGE = -6,   /* input exceeds the stated maximum of an allocation-free (static scratch) entry point */
  GMK_E_STATE = -7        /* call out of order: gmk_plan_begin while recording, gmk_plan_mark / _end while not, replay of a plan still being recorded */
};

/* /root/reference/utils.py:49-66 get_activation_function */
enum gmk_act {
  GMK_ACT_NONE = 0,
  GMK_ACT_RELU = 1,
  GMK_ACT_LEAKY_RELU = 2,
  GMK_ACT_ELU = 3,
  GMK_ACT_SELU = 4,
  GMK_ACT_GELU = 5,
  GMK_ACT_SILU = 6,
  GMK_ACT_MISH = 7,
  GMK_ACT_SIGMOID = 8 /* models/autoencoders.py:109-110 (decoder output) */
};

int gmk_version(void);
/* gmk_stream_wait_stream: everything enqueued on `waiter` after this call runs behind everything enqueued on `on`
 *   before it (hipEventRecord + hipStreamWaitEvent on an event of a ring of 64 per device, created on first use; the ring
 *   is that of the device that owns `on`, whatever the current device).  The events carry hipEventDisableSystemFence:
 *   they order streams of ONE device, whose kernels bring their own agent-scope acquire / release; they do NOT make device
 *   memory visible to the host or to another device (a default record does, by writing the L2s back and invalidating them:
 *   3 % of the train step at 25 forks and joins).  A host read or a peer access after such a wait needs a fence of its own:
 *   any stream / device synchronisation or copy of the HIP runtime, an ordinary event.
 *   No counterpart in the reference: host plumbing of the backward pass's second stream (weight gradients), which
 *   autograd's single stream does not have (main.py:51). */
int gmk_stream_wait_stream(gmk_stream_t waiter, gmk_stream_t on);
const char* gmk_strerror(int code);

/* gmk_conv_set_thin_paths / gmk_conv_get_thin_paths: which kernels run the 3x3 stride-1 layers with <= 4 channels on
 *   the image side.  thin_in: bit 0 = forward of a Cin <= 4 layer, bit 1 = data gradient of a Cout <= 4 layer on
 *   conv_thin_in.hip (else the implicit GEMM); thin_outv: 1 = forward of a Cout <= 4 layer on conv_thin_outv.hip (else
 *   conv_thin_out.hip).  A negative argument leaves that setting as it is.  Defaults: 3 and 1, or the GMK_THIN_IN /
 *   GMK_THIN_OUTV environment variables.  PER DEVICE: both calls act on the calling thread's current device, and a launch
 *   reads the word of its thread's current device (atomic: a concurrent setter yields one of the two valid choices).
 *   No counterpart in the reference: the faster choice differs between devices of the same model,
 *   so a host may time its step both ways once and keep the better one (gmk/tune.py). */
int gmk_conv_set_thin_paths(int thin_in, int thin_outv);
int gmk_conv_get_thin_paths(int* thin_in, int* thin_outv);

/* ---------------------------------------------------------------- launch plans ----------
 * The train step of /root/reference/main.py:27-53 is ~60 kernel launches and ~25 stream fork/join edges; issued one by
 * one from Python + ctypes they cost 0.8 ms of host time against 1.0 ms of GPU time at batch 256.  A plan records them
 * once and replays them from C:
 * gmk_plan_begin(&plan): from now until gmk_plan_end(), every launching entry point of this header called on THIS host
 *   thread (conv / deconv / linear / loss / reparam / pool / upsample / ew_mul / gmk_stream_wait_stream) runs as usual and
 *   is also appended to the plan: its own address and its arguments by value.  One recording per thread at a time.
 * gmk_plan_mark(): closes the current segment and returns its index (>= 0): the host side marks the points where it has
 *   work of its own between launches (a gradient bucket's all-reduce).  gmk_plan_end() closes the last segment.
 * gmk_train_step_enqueue(plan, seg_begin, seg_end, nsub, from, to, bytes): calls the recorded entry points of segments
 *   [seg_begin, seg_end) again, in order, on the recorded streams, with the recorded arguments -- except that a pointer
 *   argument inside [from[i], from[i] + bytes[i]) is moved to to[i] + (p - from[i]): that is how a step gets its new image
 *   batch and noise.  Same entry points, so same argument checks, kernel selection and results as direct calls (bit for bit).
 *   Returns the first non-zero return code, 0 otherwise.
 *   The CALLER keeps every buffer named in the plan alive and in place for as long as the plan is replayed (outputs and
 *   workspaces of the recorded step become the outputs of every replay), keeps the streams alive, and replays on the
 *   device that recorded.  The plan holds no device memory.  gmk_plan_destroy frees the host-side list. */
typedef struct gmk_plan gmk_plan;
int gmk_plan_begin(gmk_plan** plan);
int gmk_plan_mark(void);
int gmk_plan_end(void);
int gmk_plan_segments(const gmk_plan* plan);
int gmk_plan_calls(const gmk_plan* plan);
void gmk_plan_destroy(gmk_plan* plan);
int gmk_train_step_enqueue(const gmk_plan* plan, int seg_begin, int seg_end, int nsub, const void* const* from,
                           const void* const* to, const size_t* bytes);

/* ---------------------------------------------------------------- convolutions ----------
 * Implicit-GEMM fp32-MFMA kernels (v_mfma_f32_32x32x2_f32), LDS-staged input patch + weights.
 * ksize in {3,5}, padding (ksize-1)/2, stride in {1,2}.
 * Stride-1 3x3 layers with Cin % 8 == 0, Cout % 16 == 0, width <= 32 and at least 256 blocks of 128 output pixels x 64
 * channels run as Winograd F(2x2,3x3) in fp32 (forward and data gradient; same operands, same results up to fp32
 * rounding, ~3e-7 of the output scale); GMK_WINO=0 in the environment keeps the implicit GEMM everywhere.
 * Stride-1 3x3 layers with <= 4 channels on the image side and H*W >= 1024 (forward of a Cin <= 4 layer, data gradient
 * of a Cout <= 4 layer) run on the vector ALUs as packed fp32 FMAs (conv_thin_in.hip: bound by the output stream);
 * GMK_THIN_IN=0 keeps the implicit GEMM.  The forward of a layer with <= 4 output channels (Cin % 8 == 0, width 16..64)
 * runs the same way (conv_thin_outv.hip: bound by the input stream; GMK_THIN_OUTV=0: the MFMA kernel of conv_thin_out.hip).
 *
 * gmk_conv2d_fwd: y = act(conv2d(x, w) + bias); if dact != NULL also writes act'(pre-activation)
 *   (the backward pass multiplies by it instead of re-evaluating erf/exp).
 *   replaces nn.Conv2d + activation, models/autoencoders.py:33-38,46-54,82-84,98,102,106
 *   x (N,Cin,H,W)  w (Cout,Cin,k,k)  bias (Cout) or NULL  y,dact (N,Cout,H/stride,W/stride)
 */
int gmk_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* dact,
                   int N, int Cin, int H, int W, int Cout, int ksize, int stride, int act,
                   gmk_stream_t stream);
/* gmk_conv2d_dgrad: dx = conv2d_input_grad(dy, w) * dact_in  (dact_in = act' of the layer that
 *   produced x, NULL for none).  replaces autograd of nn.Conv2d (main.py:51).
 *   dy (N,Cout,H/stride,W/stride)  dx,dact_in (N,Cin,H,W) */
int gmk_conv2d_dgrad(const float* dy, const float* w, const float* dact_in, float* dx,
                     int N, int Cin, int H, int W, int Cout, int ksize, int stride,
                     gmk_stream_t stream);
/* gmk_conv2d_wgrad: dw (Cout,Cin,k,k) = weight grad, dbias (Cout) = bias grad (NULL to skip).
 *   Deterministic: per-block partial slabs in `ws`, summed in fixed order.
 *   ws_bytes >= gmk_conv2d_wgrad_workspace_bytes(...). */
size_t gmk_conv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize, int stride);
int gmk_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                     int N, int Cin, int H, int W, int Cout, int ksize, int stride,
                     void* ws, size_t ws_bytes, gmk_stream_t stream);

/* Deferred slab reduction (round 3; no counterpart in the reference: it is how autograd's per-layer weight gradients,
 *   main.py:51, are scheduled here).  gmk_conv2d_wgrad / gmk_deconv2d_wgrad with dw == NULL write only their partial slabs
 *   into `ws` (which must then be the layer's own, kept until reduced) -- for the layers where gmk_conv_wgrad_can_defer()
 *   returns 1 (3x3, the widths of the 16x16x4-MFMA kernel; others return GMK_E_UNSUPPORTED); gmk_conv_wgrad_reduce_batch
 *   then sums the slabs of up to 8 such layers in ONE launch, in the same fixed order as the per-layer reduction (bit-identical
 *   dw / dbias).  `layers`: host array; each entry repeats the layer's geometry as passed to the weight-gradient call
 *   (transposed = 1: the arguments of gmk_deconv2d_wgrad, stride 2).  dbias may be NULL. */
typedef struct gmk_wgrad_layer {
  const void* ws; float* dw; float* dbias;
  int N, Cin, H, W, Cout, ksize, stride, transposed;
} gmk_wgrad_layer;
int gmk_conv_wgrad_can_defer(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int transposed);
int gmk_conv_wgrad_reduce_batch(const gmk_wgrad_layer* layers, int n, gmk_stream_t stream);

/* Transposed convolution, stride 2, padding (k-1)/2, output_padding 1: (H,W) -> (2H,2W).
 *   replaces nn.ConvTranspose2d + activation, models/autoencoders.py:86-87,100,104
 *   x (N,Cin,H,W)  w (Cin,Cout,k,k)  y,dact (N,Cout,2H,2W) */
int gmk_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* dact,
                     int N, int Cin, int H, int W, int Cout, int ksize, int act, gmk_stream_t stream);
int gmk_deconv2d_dgrad(const float* dy, const float* w, const float* dact_in, float* dx,
                       int N, int Cin, int H, int W, int Cout, int ksize, gmk_stream_t stream);
size_t gmk_deconv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize);
int gmk_deconv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                       int N, int Cin, int H, int W, int Cout, int ksize,
                       void* ws, size_t ws_bytes, gmk_stream_t stream);

/* ---------------------------------------------------------------- linear ----------------
 * replaces nn.Linear (+ activation), models/autoencoders.py:41,57,81,95 and its autograd.
 *   x (M,K)  w (N,K)  bias (N)  y,dact (M,N).  Split-K partials (deterministic) use `ws`. */
size_t gmk_linear_workspace_bytes(int M, int K, int N);
int gmk_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* dact,
                   int M, int K, int N, int act, void* ws, size_t ws_bytes, gmk_stream_t stream);
/* dx (M,K) = dy (M,N) @ w (N,K), times dact_in (M,K) if not NULL */
int gmk_linear_dgrad(const float* dy, const float* w, const float* dact_in, float* dx,
                     int M, int K, int N, void* ws, size_t ws_bytes, gmk_stream_t stream);
/* dw (N,K) = dy^T @ x ; dbias (N) = column sums of dy (NULL to skip) */
int gmk_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias,
                     int M, int K, int N, void* ws, size_t ws_bytes, gmk_stream_t stream);

/* The VAE head and tail fused into the two fully connected layers next to the latent (what north_star calls the fused
 * tail; /root/reference/models/autoencoders.py:57-61,169-176 and main.py:36-48):
 * gmk_linear_fwd_reparam: stats (M, N = 2L) = x @ w^T + bias, interleaved (mu, log_var) as the reference's
 *   view(B, L, 2), and z (M, L) = mu + exp(log_var / 2) * eps, both written by the epilogue of the GEMM's split-K
 *   reduction (no separate reparameterisation launch).  eps (M, L).
 * gmk_linear_dgrad_reparam: dz = dy (M, N) @ w (N, K = L) is never stored: the reduction's epilogue writes
 *   dstats[.,.,0] = dz + kl_scale * dKL/dmu, dstats[.,.,1] = dz * eps * sigma / 2 + kl_scale * dKL/dlog_var
 *   (kl_scale = beta / batch; the --mse prior's gradient when mse_prior != 0), i.e. autograd of the
 *   reparameterisation plus the gradient of beta * KL; pair it with GMK_LOSS_NO_DSTATS below.
 *   Workspaces as for gmk_linear_fwd / gmk_linear_dgrad. */
int gmk_linear_fwd_reparam(const float* x, const float* w, const float* bias, const float* eps, float* stats,
                           float* z, int M, int K, int N, void* ws, size_t ws_bytes, gmk_stream_t stream);
int gmk_linear_dgrad_reparam(const float* dy, const float* w, const float* stats, const float* eps, float* dstats,
                             int M, int K, int N, float kl_scale, int mse_prior, void* ws, size_t ws_bytes,
                             gmk_stream_t stream);

/* ---------------------------------------------------------------- VAE glue --------------
 * gmk_reparam_fwd: z[b,l] = stats[b,l,0] + exp(0.5*stats[b,l,1]) * eps[b,l]
 *   replaces models/autoencoders.py:169-176.  stats (B,L,2) interleaved mu/log_var. */
int gmk_reparam_fwd(const float* stats, const float* eps, float* z, int64_t B, int64_t L,
                    gmk_stream_t stream);
/* gmk_reparam_bwd: dstats[b,l,0] += dz ; dstats[b,l,1] += dz*eps*0.5*exp(0.5*log_var)
 *   (accumulates onto the KL gradient written by gmk_vae_loss_fwd_bwd; if accumulate==0 it
 *   overwrites).  autograd of models/autoencoders.py:171-176. */
int gmk_reparam_bwd(const float* dz, const float* stats, const float* eps, float* dstats,
                    int64_t B, int64_t L, int accumulate, gmk_stream_t stream);

/* gmk_vae_loss_fwd_bwd: ONE coalesced HBM-bound kernel (+ a 1-block finalize) computing
 *   recon = sigmoid(pre) (if GMK_LOSS_SIGMOID, models/autoencoders.py:109-110),
 *   rec  = mean((recon-image)^2)                         (main.py:31),
 *   kl   = mean_b(-0.5*sum_l(1+lv-mu^2-exp(lv)))         (main.py:45)
 *          or sum_l mu^2 + sum_l (sigma-1)^2 if GMK_LOSS_MSE_PRIOR (main.py:39-42),
 *   total = rec + beta*kl                                 (main.py:47-48),
 *   and their gradients: dpre = d total/d pre, dstats = beta * d kl/d stats.
 *   pre,image,dpre,recon_out: (B, chw) ; stats,dstats: (B,L,2) (NULL when not variational).
 *   recon_out may be NULL.  losses: 3 floats on device {rec, kl, total}.
 *   ws: >= gmk_vae_loss_workspace_bytes() bytes. */
/* GMK_LOSS_NO_DSTATS: the KL terms are summed but no dstats is written: the caller applies the KL gradient together
 *   with the reparameterisation backward in gmk_linear_dgrad_reparam (the fused train step does). */
enum { GMK_LOSS_SIGMOID = 1, GMK_LOSS_MSE_PRIOR = 2, GMK_LOSS_NO_DSTATS = 4 };
size_t gmk_vae_loss_workspace_bytes(void);
int gmk_vae_loss_fwd_bwd(const float* pre, const float* image, const float* stats,
                         float* recon_out, float* dpre, float* dstats, float* losses,
                         int64_t B, int64_t chw, int64_t L, float beta, int flags,
                         void* ws, size_t ws_bytes, gmk_stream_t stream);
/* gmk_vae_loss_fwd_bwd_fused: the same in ONE launch (what the fused train step uses): the block that finishes last
 *   sums the partials (same fixed order, same results).  `sync`: 256 bytes of device memory, 256-byte aligned, ZERO
 *   before the first call and used by nothing else; the kernel leaves it zero.  Calls that share `ws` / `sync`
 *   must be stream-ordered. */
int gmk_vae_loss_fwd_bwd_fused(const float* pre, const float* image, const float* stats,
                               float* recon_out, float* dpre, float* dstats, float* losses,
                               int64_t B, int64_t chw, int64_t L, float beta, int flags,
                               void* ws, size_t ws_bytes, void* sync, gmk_stream_t stream);
/* gmk_vae_loss_eval: the no-grad form of the same kernel, for the eval pass of
 *   /root/reference/main.py:73-111 (compute_test_loss): no gradient stores; out5 = 5 floats on device
 *   {rec, kl, total = rec + beta*kl, mean_{b,l} |mu| (main.py:101), mean_{b,l} exp(log_var/2) (main.py:102)};
 *   the last two are 0 when stats == NULL.  recon_out may be NULL.  Same workspace as above. */
int gmk_vae_loss_eval(const float* pre, const float* image, const float* stats, float* recon_out,
                      float* out5, int64_t B, int64_t chw, int64_t L, float beta, int flags,
                      void* ws, size_t ws_bytes, gmk_stream_t stream);

/* gmk_vae_loss_fwd_bwd_static: gmk_vae_loss_fwd_bwd without the workspace argument.  NO COUNTERPART IN THE
 *   REFERENCE: it exists for callers that cannot allocate or carry a workspace -- code under HIP graph capture,
 *   tight serving loops, harnesses that forbid hipMalloc.  The block partials live in a 32 KiB file-scope
 *   __device__ array of the library (one copy per GPU, held until the process ends); the call allocates nothing
 *   and frees nothing.  Same kernels and therefore bit-identical outputs.  Limits: B*chw and B*L at most
 *   GMK_LOSS_STATIC_MAX_ELEMS (2^28) each, else GMK_E_TOO_LARGE.  The first call on a device resolves the array's
 *   device address (hipGetSymbolAddress): make it outside graph capture, with that device current.  All calls on
 *   one GPU share the array: calls on one stream are ordered, calls that overlap on different streams race. */
#define GMK_LOSS_STATIC_MAX_ELEMS (1ll << 28)
int gmk_vae_loss_fwd_bwd_static(const float* pre, const float* image, const float* stats,
                                float* recon_out, float* dpre, float* dstats, float* losses,
                                int64_t B, int64_t chw, int64_t L, float beta, int flags,
                                gmk_stream_t stream);

/* ---------------------------------------------------------------- pool / upsample -------
 * Pool + upsample variants of the autoencoder (models/autoencoders.py:40,48,52,89,100,104).
 * `planes` = N*C.  maxpool: (planes,H,W) -> (planes,H/2,W/2), idx (uint8, 0..3) records the
 * arg-max for the backward pass (first maximum in scan order, like PyTorch).
 * upsample x2: nn.Upsample(scale_factor=2, mode) with align_corners=False.
 * The *_bwd kernels multiply by dact_in (act' of the producing layer) when it is not NULL. */
enum gmk_upsample_mode { GMK_UP_NEAREST = 0, GMK_UP_BILINEAR = 1, GMK_UP_BICUBIC = 2 };
int gmk_maxpool2_fwd(const float* x, float* y, unsigned char* idx, int64_t planes, int H, int W,
                     gmk_stream_t stream);
int gmk_maxpool2_bwd(const float* dy, const unsigned char* idx, const float* dact_in, float* dx,
                     int64_t planes, int H, int W, gmk_stream_t stream);
int gmk_upsample2_fwd(const float* x, float* y, int64_t planes, int H, int W, int mode,
                      gmk_stream_t stream);
int gmk_upsample2_bwd(const float* dy, const float* dact_in, float* dx, int64_t planes, int H, int W,
                      int mode, gmk_stream_t stream);
/* out = a * b elementwise (autograd glue: d_pre = d_recon * sigmoid') */
int gmk_ew_mul(const float* a, const float* b, float* out, int64_t n, gmk_stream_t stream);

/* ---------------------------------------------------------------- optimizer -------------
 * gmk_adamw_step: one launch over a flat buffer holding every parameter (n % 4 == 0, padded):
 *   torch.optim.AdamW(lr, betas, eps, weight_decay) single step, `step` = 1-based step index;
 *   the gradient is multiplied by gscale first (1/world_size after a SUM all-reduce).
 *   replaces optim.step(), /root/reference/main.py:52,214. */
int gmk_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, float gscale,
                   gmk_stream_t stream);

/* ---------------------------------------------------------------- data feeder -----------
 * gmk_cifar_batch: out[b] = ToTensor(data[index[b]]) (uint8 -> fp32 / 255), and when `augment` != 0 the reference's
 *   train transform RandomCrop(H, padding=pad) + RandomHorizontalFlip on top: per sample b,
 *   h = mix32(mix32(b * 0x9E3779B1 + lo32(seed)) + hi32(seed) * 0x85EBCA6B)  (mix32 = lowbias32),
 *   dy = h % (2 pad + 1), dx = (h >> 8) % (2 pad + 1), flip = (h >> 16) & 1,
 *   out[b][c][y][x] = data[index[b]][c][y + dy - pad][(flip ? W-1-x : x) + dx - pad]  (0 outside the image).
 *   replaces torchvision's CIFAR10 + transforms + DataLoader, /root/reference/dataloaders.py:23-41 (the dataset
 *   stays resident in HBM as uint8; the host only supplies the shuffled index and a seed per batch).
 *   data (n,C,H,W) uint8, index (B) int64, out (B,C,H,W) fp32; W % 4 == 0. */
int gmk_cifar_batch(const unsigned char* data, const int64_t* index, float* out, int64_t B, int C, int H, int W,
                    int pad, int augment, uint64_t seed, gmk_stream_t stream);

/* ---------------------------------------------------------------- FID / inception stats --
 * gmk_gram_accumulate: n fp32 feature rows X (n, D) -> G (D,D) += X^T X, s (D) += column sums, both
 *   fp64, on v_mfma_f64_16x16x4_f64 (products of fp32 values are exact in fp64, so the moments do
 *   not depend on how samples are sharded over calls or GPUs beyond ~1e-16 relative).
 *   replaces np.cov / mean of /root/reference/metrics.py:75-79 (moments are combined across ranks
 *   with ONE all-reduce of (n, s, G) by the host side).  G, s must be zeroed by the caller first. */
size_t gmk_gram_workspace_bytes(int64_t n, int D);
int gmk_gram_accumulate(const float* X, int64_t n, int D, double* G, double* s, void* ws, size_t ws_bytes,
                        gmk_stream_t stream);
/* gmk_fid_from_moments: mu = s/n, Sigma = (G - s s^T/n)/(n-1) + diag_eps*I for both sets, then
 *   fid = |mu1-mu2|^2 + tr S1 + tr S2 - 2 tr sqrt(S1 S2), with tr sqrt(S1 S2) = tr sqrt(S1^1/2 S2 S1^1/2)
 *   from two symmetric coupled Newton-Schulz square roots (`iters` iterations each, relative ridge
 *   `delta`), fp64 MFMA GEMMs, no host synchronisation.
 *   replaces scipy.linalg.sqrtm + the trace formula of metrics.py:90-108.
 *   out: 8 doubles on device {dmu2, tr1, tr2, fro, c, tr_sqrt, fid, -}. */
size_t gmk_fid_workspace_bytes(int D);
int gmk_fid_from_moments(const double* G1, const double* s1, int64_t n1, const double* G2, const double* s2,
                         int64_t n2, int D, int iters, double delta, double diag_eps, double* out,
                         void* ws, size_t ws_bytes, gmk_stream_t stream);
/* gmk_softmax_stats: logits (n, C<=32) -> out (2C+4 doubles, ACCUMULATED; zero first):
 *   [0,C) sum p ; [C,2C) arg-max histogram ; [2C] sum_n -sum_c p log(p+eps) ; [2C+1] sum top-1 ;
 *   [2C+2] sum top-N ; [2C+3] n.   replaces metrics.py:144-169 and :231-253 (one all-reduce of `out`
 *   combines ranks). */
int gmk_softmax_stats(const float* logits, int64_t n, int C, int topn, float eps, double* out,
                      gmk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GMK_H */
