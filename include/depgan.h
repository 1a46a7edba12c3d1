/* libdepgan -- C ABI of the MI355X-native DEP-GAN two-critic WGAN-GP hot path.
 *
 * The reference (febrianrachmadi/dep-gan-im) has no FFI: its boundary is the
 * slice of the Keras API that DEP-GAN_PROB_IM_twoCritics_training_4fold.py
 * ("GT") touches.  Each entry point below names the reference construct it
 * replaces.  All pointers are plain device (HIP) or host pointers, no
 * framework types.  Every function returns 0 on success; on failure
 * depgan_last_error() describes what went wrong.  A context is not thread safe;
 * independent contexts (one per rank / GPU) are.
 *
 * Tensors are NHWC fp32, Keras weight layouts (Conv2D HWIO, Conv2DTranspose
 * (kh,kw,Cout,Cin), Dense (in,out)).
 */
#ifndef DEPGAN_H
#define DEPGAN_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct depgan_ctx depgan_ctx;

/* ABI guard: bump when depgan_config or the meaning of an entry point changes.  depgan_create rejects a
 * depgan_config whose struct_size is not sizeof(depgan_config) of THIS header (a caller compiled or bound against an
 * older layout would otherwise make the library read past its struct). */
#define DEPGAN_ABI_VERSION 3
int depgan_abi_version(void);
size_t depgan_config_size(void);
/* first 32 hex digits of the sha256 over the sources this binary was built from (dep_gan_im_amd/build.py::source_hash):
 * the Python binding compares it with the sources lying next to it and refuses a stale library */
const char* depgan_source_hash(void);

typedef struct depgan_config {
  int struct_size;  /* = sizeof(depgan_config); checked by depgan_create                */
  int batch;        /* per-device batch size (batchSize, GT:42)                       */
  int height;       /* imageSize (GT:40); must be a multiple of 16                    */
  int width;
  int nicg;         /* generator input channels (GT:22)                               */
  int first_fm;     /* first_fm_G (GT:35); 32                                          */
  float im_thresh;  /* IM_TRSH (GT:25-29)                                              */
  float delta;      /* WGAN-GP weight (GT:37)                                          */
  float lrD, lrG;   /* GT:44-45                                                        */
  float beta1, beta2, adam_eps; /* Adam(beta_1=0, beta_2=0.9), K.epsilon() (GT:549)    */
  int nc_out;       /* generator head channels: 1 = DEP-GAN (tanh, GT:520); 4 = DEP-UResNet (softmax, UT:583).
                       0 is read as 1.                                                    */
  int bf16_weights; /* BASELINE config 4: 1 = every "/kernel" tensor is rounded to bf16 (RNE) before use, products
                       accumulate in fp32, the fp32 master copy and the Adam state stay fp32; 0 = fp32 weights */
  int bf16_mfma;    /* BASELINE config 4 on the bf16 matrix pipe (needs bf16_weights = 1): the MFMA convolutions AND the
                       weight-gradient contractions round their operands to bf16 (RNE) while staging them and run
                       v_mfma_f32_32x32x16_bf16 with fp32 accumulation; everything between them stays fp32.
                       0 = fp32 matrix pipe  */
  int f32_split;    /* 0 (default): fp32 products on v_mfma_f32_32x32x2_f32.  6 or 3 (opt-in, never the benchmark's
                       headline; excludes bf16_weights / bf16_mfma): every fp32 operand of the MFMA convolutions is split
                       exactly into three (two) bf16 terms and the six (three) largest cross products run on
                       v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- dropped terms are below 2^-24 (2^-16) of
                       |x||w|, i.e. fp32-grade (6) or TF32-grade-plus (3) products at 2.7x (5.3x) the fp32 matrix rate */
} depgan_config;

enum { DEPGAN_NET_G = 0, DEPGAN_NET_D_Y2 = 1, DEPGAN_NET_D_DEM = 2 };
enum { DEPGAN_ARENA_PARAMS = 0, DEPGAN_ARENA_NONTRAINABLE = 1, DEPGAN_ARENA_GRADS = 2,
       DEPGAN_ARENA_ADAM_M = 3, DEPGAN_ARENA_ADAM_V = 4 };

const char* depgan_last_error(void);

/* Gen_UNet2D(...) + 2 x Dis_C2D_FCN1(...) + the loss graph of GT:513-598.
 * Weights start zeroed: load them with depgan_arena_ptr + depgan_weights_changed. */
int depgan_create(const depgan_config* cfg, depgan_ctx** out);
void depgan_destroy(depgan_ctx* ctx);
/* all work is enqueued on this hipStream_t (default: the null stream) */
int depgan_set_stream(depgan_ctx* ctx, void* hip_stream);

/* Data parallelism (the reference is single-GPU, GT:13).  The library stays free of any communication
 * dependency: the host registers ONE function that all-reduces (sum) n device floats in place across the ranks,
 * ENQUEUED on hip_stream (RCCL via torch.distributed in dep_gan_im_amd/dist.py); it must not synchronise the host.
 * With a hook registered every *_step / depgan_gen_iteration / depgan_g_eval_multi call all-reduces, per network
 * update, that network's gradient arena together with the un-normalised loss pieces (one message), divides the
 * gradient by `world` inside Adam and reports GLOBAL scalars, identical on every rank (hence the same best-of-k
 * noise choice).  *_grads calls never communicate.  fn == NULL removes the hook; world = 1 with a hook is a one-rank
 * job (the hook is still called). */
typedef int (*depgan_allreduce_fn)(void* user, float* dev_ptr, long n, void* hip_stream);
int depgan_set_allreduce(depgan_ctx* ctx, depgan_allreduce_fn fn, void* user, int world);

/* Direct RCCL binding ("ncclAllReduce ... one call per network per step").  The library calls
 * ncclAllReduce(arena, nTrain + 8, ncclFloat, ncclSum, comm, stream) itself, on the context's stream, for every network
 * update (and once per best-of-k evaluation block); librccl is resolved at run time -- the copy already loaded in the
 * process (PyTorch-ROCm ships one), else dlopen("librccl.so.1") -- so libdepgan.so has no link dependency on it.
 *   depgan_rccl_unique_id:  rank 0 creates the 128-byte ncclUniqueId; the host hands it to every rank by any means
 *                           (dep_gan_im_amd/dist.py: one torch.distributed broadcast of the bytes).
 *   depgan_rccl_init:       collective over all ranks: ncclCommInitRank on the context's device.  Replaces a hook
 *                           registered with depgan_set_allreduce; from then on the closures behave as described there
 *                           (global scalars, gradients divided by `world` inside Adam).  world = 1 is a one-rank job.
 *   depgan_rccl_broadcast:  ncclBroadcast of n device floats from `root`, enqueued on the context's stream (replica
 *                           initialisation: rank 0's weights, BN statistics and Adam state).
 *   depgan_rccl_info:       communicator size and rank as RCCL reports them (ncclCommCount / ncclCommUserRank) and
 *                           the number of collectives this context has issued.
 *   depgan_rccl_shutdown:   ncclCommDestroy (depgan_destroy does it as well). */
#define DEPGAN_RCCL_ID_BYTES 128
int depgan_rccl_unique_id(void* id_out);
int depgan_rccl_init(depgan_ctx* ctx, const void* id, int rank, int world);
int depgan_rccl_broadcast(depgan_ctx* ctx, float* dev_ptr, long n, int root);
int depgan_rccl_info(depgan_ctx* ctx, int* nranks, int* rank, long* collectives_issued);
int depgan_rccl_shutdown(depgan_ctx* ctx);

/* model.trainable_weights / get_weights / set_weights (GT:549, 892; GE:383) */
int depgan_param_count(depgan_ctx* ctx, int net);
int depgan_param_info(depgan_ctx* ctx, int net, int index, char* name, int name_cap, int shape[4], int* ndim,
                      long* offset, int* trainable);
long depgan_arena_floats(depgan_ctx* ctx, int net, int arena);
float* depgan_arena_ptr(depgan_ctx* ctx, int net, int arena); /* device pointer */
/* call after writing into a PARAMS / NONTRAINABLE arena from outside */
int depgan_weights_changed(depgan_ctx* ctx, int net);

/* Model.predict (GT:846-848, 859; GE:621): n <= batch samples, learning phase 0 */
int depgan_g_forward(depgan_ctx* ctx, const float* x_dev, const float* z_dev, float* out_dev, int n);
int depgan_d_forward(depgan_ctx* ctx, int net, const float* img_dev, float* out_dev, int n);

/* netD_y2_train / netD_dem_train ([y2, x, z, ep] -> [loss_real, loss_fake]; GT:550-552, 569-571).
 * *_grads leaves d loss / d theta_D in the GRADS arena without updating (so a
 * data-parallel caller can all-reduce it), depgan_apply_adam applies
 * Adam.get_updates (GT:549, 568, 594); *_step does both. */
int depgan_critic_grads(depgan_ctx* ctx, int net, const float* y2_dev, const float* x_dev, const float* z_dev,
                        const float* ep_dev, float out_host[2]);
int depgan_critic_step(depgan_ctx* ctx, int net, const float* y2_dev, const float* x_dev, const float* z_dev,
                       const float* ep_dev, float out_host[2]);
/* netG_no_update / netG_train ([x, y2, z] -> [loss, loss_fake, loss_fake_dem, M1, M3, M4]; GT:595-598) */
int depgan_g_eval(depgan_ctx* ctx, const float* x_dev, const float* y2_dev, const float* z_dev, float out_host[6]);
int depgan_g_grads(depgan_ctx* ctx, const float* x_dev, const float* y2_dev, const float* z_dev, float out_host[6]);
/* The best-of-k noise search of the driver (GT:868-877: k = 10 calls of netG_no_update on the SAME batch with k
 * noises, then argmin of the total loss): k evaluations enqueued back to back, one host synchronisation.
 * z_all_dev: (k, batch, 32) ; out_host: k x 6 scalars as depgan_g_eval ; sums_host (may be NULL): k x 8
 * un-normalised pieces as depgan_last_sums (for the data-parallel combine).  1 <= k <= DEPGAN_MAX_MULTI. */
#define DEPGAN_MAX_MULTI 32
int depgan_g_eval_multi(depgan_ctx* ctx, const float* x_dev, const float* y2_dev, const float* z_all_dev, int k,
                        float* out_host, float* sums_host);
int depgan_g_step(depgan_ctx* ctx, const float* x_dev, const float* y2_dev, const float* z_dev, float out_host[6]);
int depgan_apply_adam(depgan_ctx* ctx, int net);
/* Adam `iterations` of a network's optimiser (GT:549, 568, 594), for checkpoint / resume */
long depgan_get_adam_step(depgan_ctx* ctx, int net);
int depgan_set_adam_step(depgan_ctx* ctx, int net, long t);

/* One generator iteration of the reference schedule (GT:791-829, 868-878) with ONE host synchronisation:
 *   n_y2  critic-Y2 updates  on the batches  x_y2 + j*stride, y2_y2 + j*stride_y   (j = 0 .. n_y2-1)   GT:802-814
 *   n_dem critic-DEM updates on the batches  x_dem + j*stride, ...                                        GT:817-829
 *   k evaluations of the generator loss on (x_gen, y2_gen) with the noises z_gen[0..k)                    GT:868-874
 *   arg-min of the total loss (first minimum of the float32 values, as np.argmin)                         GT:875-876
 *   one generator update with that noise                                                                  GT:878
 * everything enqueued back to back (the arg-min and the noise gather run on the device), all scalars fetched at the
 * end.  batch_stride = samples between the starts of consecutive batches (batch for data resident in HBM as one
 * array; world*batch when the rank takes every world-th batch).  z_*: (n, batch, 32), ep_*: (n, batch).
 * out_host: n_y2 x 2 [loss_real, loss_fake], then n_dem x 2, then k x 6, then the 6 scalars of the update
 * (2 n_y2 + 2 n_dem + 6 k + 6 floats); *best_host = chosen noise index.  n_y2, n_dem >= 0, 1 <= k <= DEPGAN_MAX_MULTI;
 * n_y2 + n_dem <= DEPGAN_MAX_CRITIC_STEPS. */
#define DEPGAN_MAX_CRITIC_STEPS 256
int depgan_gen_iteration(depgan_ctx* ctx, const float* x_y2, const float* y2_y2, const float* z_y2, const float* ep_y2,
                         int n_y2, const float* x_dem, const float* y2_dem, const float* z_dem, const float* ep_dem,
                         int n_dem, long batch_stride, const float* x_gen, const float* y2_gen, const float* z_gen, int k,
                         float* out_host, int* best_host);

/* DEP-UResNet supervised path (DEP-UResNet-wNoises-training-4fold.py "UT"; contexts created with nc_out = 4):
 * my_network.fit / train_on_batch (UT:427, 602-606) = learning phase 1: batch-statistics BatchNorm with
 * moving-average updates, Dropout(0.25) after conv_10 (UT:388; drop_seed 0 disables it), softmax +
 * categorical cross-entropy, Adam(beta1, beta2 of the config).  labels: one-hot (n,H,W,4) fp32.
 * n: samples in this call (1..batch; the last batch of a keras epoch may be short); loss_host: mean loss.
 * depgan_uresnet_grads leaves the gradients in the G arena and, like any phase-1 forward pass, moves
 * the BN moving statistics; depgan_uresnet_step also applies Adam.                                   */
int depgan_uresnet_grads(depgan_ctx* ctx, const float* x_dev, const float* z_dev, const float* labels_dev, int n,
                         unsigned drop_seed, float* loss_host);
int depgan_uresnet_step(depgan_ctx* ctx, const float* x_dev, const float* z_dev, const float* labels_dev, int n,
                        unsigned drop_seed, float* loss_host);
/* validation loss in learning phase 0 (UT:606); n in 1..batch */
int depgan_uresnet_eval(depgan_ctx* ctx, const float* x_dev, const float* z_dev, const float* labels_dev, int n,
                        float* loss_host);

/* Un-normalised pieces of the last critic / generator evaluation, for exact
 * data-parallel reporting: critic: [sum D(real), sum D(fake), sum (norm-1)^2, n];
 * generator: [sum D_y2(fake), sum D_dem(attr), sum |attr-real_dem|, sum wr, sum wf, sum wr*wf, n, n*H*W]. */
int depgan_last_sums(depgan_ctx* ctx, float out_host[8]);

/* per-kernel-class device timing (HIP events on the context stream) */
int depgan_profile_enable(depgan_ctx* ctx, int on);
/* class 0: MFMA conv (fwd / bwd-data / u-forward), 1: MFMA wgrad, 2: everything else */
int depgan_profile_read(depgan_ctx* ctx, int klass, double* total_ms, long* launches, double* flops);
/* sum over the class's recorded launches of the algorithmic HBM bytes (operands once, results once; class 0 only) */
int depgan_profile_read_bytes(depgan_ctx* ctx, int klass, double* bytes);
int depgan_profile_reset(depgan_ctx* ctx);
/* one CSV row per recorded launch: class,label,ms,gflop,mbytes,kernel (algorithmic GFLOP / MB of the launch; kernel =
 * the template instantiation as rocprofv3 names it, for the MFMA convolution class) */
int depgan_profile_dump(depgan_ctx* ctx, const char* path);

/* ---- evaluation step after the path (DEP-GAN_testing_4fold.py "GE":616-807) ----
 * Stateless; device pointers; work is enqueued on `stream`.  Number types follow the reference's NumPy statements:
 * depgan_eval_accumulate: acc += (double)(pred * mask) (mask may be NULL) -- the float64 running sum (np.zeros,
 *   GE:617) of the n_repeat float32 masked predictions (GE:618-624);  depgan_eval_divide: acc /= divisor in float64
 *   (GE:628: output_img_pred_mean / float(n_repeat)).
 * depgan_eval_counts: the integer census behind the volumes and the six Dice figures (GE:637-790):
 *   x (npix, nicg) float32 input maps, pred (npix) FLOAT64 mean predicted DEM; optional (NULL = absent) code_real
 *   (npix, values 0..3), mask1 / wmh1 / mask2 / wmh2 / prob2 (npix).  fake = clip(x0 + pred, -1, 1) and its
 *   comparisons run in float64 against thr; the float32 arrays x and prob2 are compared against (float)thr, as NumPy
 *   does for a float32 array and a Python float.  out_host:
 *   [0] nnz(mask1*wmh1) [1] nnz(mask2*wmh2) [2] #(x >= thr) [3] #(prob2 >= thr) [4] nnz(mask2*[fake > thr]);
 *   change code of the prediction 1 shrink / 2 grow / 3 stay; then triples
 *   (#both, #real, #fake) for code 1, 2, 3 at [5..13], for code > 0 at [14..16], for code in {1,2} at [17..19]. */
#define DEPGAN_EVAL_NCOUNT 20
int depgan_eval_accumulate(const float* pred_dev, const float* mask_dev, double* acc_dev, long n, void* stream);
int depgan_eval_divide(double* acc_dev, long n, double divisor, void* stream);
int depgan_eval_counts(const float* x_dev, int nicg, const double* pred_dev, const float* code_real_dev,
                       const float* mask1_dev, const float* wmh1_dev, const float* mask2_dev, const float* wmh2_dev,
                       const float* prob2_dev, long npix, double thr, long long out_host[DEPGAN_EVAL_NCOUNT],
                       void* stream);

/* ---- data step in front of the path (DEP-GAN_PROB_IM_twoCritics_training_4fold.py "GT": 93-118 load_data /
 * data_prep, 124-146 map_image_to_intensity_range, 667-723 masking / clamping / channel concat) ----
 * One subject: volumes are device fp32 arrays in NIfTI file order (x fastest: element (x,y,z) at x + X*(y + Y*z)),
 * already cast to float32 as data_prep does.  Writes the training slices x_out (Z, X, Y, nicg) and y2_out (Z, X, Y, 1):
 *   prob_1 = p1*icv1 [*(1 - sl1)] clamped at 0;  flair_1 = f1*icv1 [*(1 - sl1)] mapped to [0,1] by the subject's min /
 *   max (percentile 0);  prob_2 = p2*icv2 [*(1 - sl2)] clamped at 0.  sl1 / sl2 may be NULL (no stroke-lesion mask:
 *   GT:691, 699 skip it when the file is missing); f1 may be NULL when nicg = 1.  Bit-identical to the NumPy statements.
 *   scratch: depgan_data_prep_scratch_floats(X, Y, Z) device floats (needed for nicg = 2).  Enqueued on `stream`. */
size_t depgan_data_prep_scratch_floats(int X, int Y, int Z);
int depgan_data_prep_subject(const float* p1_dev, const float* f1_dev, const float* icv1_dev, const float* sl1_dev,
                             const float* p2_dev, const float* icv2_dev, const float* sl2_dev, int X, int Y, int Z,
                             int nicg, float* x_out_dev, float* y2_out_dev, float* scratch_dev, void* stream);

/* ---- parity-test surface: the tensors a training closure left behind ----
 * The step functions are piecewise linear in the ReLU signs and max-pool arg-maxes of the forward passes (GT:256-309
 * activations, GT:322-335 pools; the gradient penalty of GT:543-549 differentiates through them twice).  A parity
 * test that wants a bound EVERY evaluation must meet compares the gradients with a float64 restatement evaluated
 * under the very masks this library used; these two calls hand them out.
 * depgan_debug_capture(ctx, 1): from now on every critic closure keeps a copy of the post-ReLU activations of its
 *   interpolated ("mixed", GT:538 / 557) pass, which the penalty's second pass otherwise overwrites in place.
 * depgan_debug_tensor: copies one internal tensor to host memory as a dense (N, H, W, C) float32 array and reports
 *   its shape; host_dst == NULL only reports the shape.  Names (layer names as in the reference, GT:256-309, 319-339):
 *     "g/out/<layer>"   output of a generator trunk layer of the last generator pass: conv / FiLM block / deconv
 *                       (post-ReLU), "skip1..3" (the pooled tensor), "gen_segmentation" (tanh output)
 *     "g/u/<layer>"     BatchNorm output of a FiLM block's convolution (GT:402; kept by training passes only)
 *     "g/heads"         the 14 noise-MLP head outputs, (N, 1, 1, 1024) in creation order (GT:363-395)
 *     "g/noise_a0", "g/noise_a1"   post-ReLU trunk activations of the noise MLP (GT:358-359), (N, 1, 1, 1024)
 *     "d/act/<layer>"   post-ReLU activations of critic layer dis_0a .. dis_8 of the last critic passes, 3*batch
 *                       sample slots [real | fake | mixed] (a generator pass leaves D_y2(fake_y2) in slots [0, batch) and
 *                       D_dem(attr) in [batch, 2 batch))
 *     "d/mixed/<layer>" the captured copy of the mixed pass (batch samples; needs depgan_debug_capture)
 * Status 1 for an unknown name, a tensor that was not captured, or cap_floats too small. */
int depgan_debug_capture(depgan_ctx* ctx, int on);
int depgan_debug_tensor(depgan_ctx* ctx, const char* name, float* host_dst, long cap_floats, int shape[4]);

/* ---- single operators (unit-test surface; device pointers) ---- */
/* path: 0 auto, 1 fp32 MFMA implicit GEMM, 2 direct, 3 bf16 MFMA implicit GEMM (both operands rounded to bf16, RNE),
 * 4 / 5: fp32 operands split into 2 / 3 bf16 terms, 3 / 6 products on the bf16 pipe (depgan_config.f32_split = 3 / 6),
 * 6: fp32 MFMA with 8-channel chunks, workgroup tiles (large 3x3 launches with more than 64 input channels),
 * 7: the wave-private form of 6 (csrc/igemm_wp.hip: large 3x3 launches with Cin <= 64; bit-identical to 6),
 * 8: Winograd F(2x2,3x3) on the fp32 matrix pipe (csrc/igemm_wino.hip: 3x3, Cin % 8 == 0, Cout % 32 == 0, even H, W) */
int depgan_op_conv2d(const float* in, const float* w_hwio, const float* bias, float* out, int B, int H, int W,
                     int Cin, int Cout, int KS, int relu, int path, void* hip_stream);
int depgan_op_conv2d_bwd_data(const float* dy, const float* w_hwio, float* dx, int B, int H, int W, int Cin,
                              int Cout, int KS, int path, void* hip_stream);
int depgan_op_conv2d_wgrad(const float* x, const float* dy, float* dw_hwio, int B, int H, int W, int Cin, int Cout,
                           int KS, void* hip_stream);
/* the same contraction on the bf16 matrix pipe (depgan_config.bf16_mfma: both operands rounded to bf16, RNE, while
 * staged; fp32 accumulation).  KS in {1, 3, 5}, Cin >= 8, Cin and Cout multiples of 4; status 3 otherwise. */
int depgan_op_conv2d_wgrad_bf16(const float* x, const float* dy, float* dw_hwio, int B, int H, int W, int Cin,
                                int Cout, int KS, void* hip_stream);

/* diagnostics: MFMA conv with per-workgroup phase stamps (16 x u64 per workgroup: start, after first prefetch
 * issue, stage-0 ready, stage-1 ready, MFMAs done, end, realtime ticks, HW_ID) */
int depgan_op_conv2d_stamps(const float* in, const float* w_hwio, float* out, int B, int H, int W, int Cin, int Cout,
                            int KS, unsigned long long* stamps, int reps, void* hip_stream);
int depgan_op_maxpool(const float* in, float* out, int B, int Ho, int Wo, int C, void* hip_stream);
/* Forward of the 2x2 / stride-2 transposed convolution (Conv2DTranspose, GT:308) on the fused four-tap kernel:
 * out[b][2i+di][2j+dj][co] = act((sum_ci in[b][i][j][ci] w[di][dj][co][ci] + bias[co]) * scale[co] + shift[co]).
 * w_hwoi is the Keras kernel (2, 2, Cout, Cin); bias / scale+shift may be null; out is dense (B, 2H, 2W, Cout).
 * Status 3 when the kernel does not cover the shape (Cin in {64, 96, 128}, Cout % 32 == 0, H and W
 * powers of two, W >= 8, B H W % 32 == 0). */
int depgan_op_deconv2x2(const float* in, const float* w_hwoi, const float* bias, const float* scale,
                        const float* shift, float* out, int B, int H, int W, int Cin, int Cout, int relu,
                        void* hip_stream);

/* Weight gradient of the same layer on the fused kernel (four taps + the column sums of dout in one launch):
 * dw_hwoi[di][dj][co][ci] = sum_p dout[b][2i+di][2j+dj][co] in[b][i][j][ci]; colsum[co] (optional) = sum of dout over
 * all output pixels.  Status 3 when not covered (Cin = Cout in {64, 96, 128}, H and W powers of two). */
int depgan_op_deconv2x2_wgrad(const float* in, const float* dout, float* dw_hwoi, float* colsum, int B, int H, int W,
                              int Cin, int Cout, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
