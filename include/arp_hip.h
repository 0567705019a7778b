/* arp_hip.h -- C ABI of libarp_hip.so: MI355X (gfx950) hot paths of ARP-DT.
 *
 * The reference (csmile-1006/ARP) is pure Python and has no FFI; the two hot paths are Python
 * closures.  Each entry point below names the reference seam it replaces (paths relative to
 * /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions: every function returns 0 on success and a negative value on error; the message is
 * available from arp_last_error() (thread-local).  The caller owns every host buffer.  The library
 * owns device memory behind opaque handles.  A handle is bound to one HIP device and one HIP
 * stream and is NOT thread-safe (one host thread / process per GPU, as the reference's
 * single-threaded callers).  Calls are synchronous on return unless the name ends in _async.
 * There is NO CPU fallback: without a usable GPU every compute entry point fails.
 */
#ifndef ARP_HIP_H
#define ARP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARP_MODE_F32 0  /* parity mode: f32 storage, f32-input MFMA (exact f32 FMA chains)        */
#define ARP_MODE_BF16 1 /* throughput mode: bf16 GEMM operands, f32 accumulate/residual/LN/softmax */
#define ARP_MODE_F16 2  /* IEEE binary16 GEMM operands (same MFMA rate as bf16, 11 significand bits -- what openai/CLIP itself runs on a
                           GPU), everything else as BF16.  The DEFAULT throughput mode of every handle
                           (arp_clip, arp_dt, arp_ft, arp_enc): the 16-bit mode that meets north_star's 1e-4 / 1e-3; the train steps
                           carry an exact power-of-two scale on their backward activations (binary16's exponent range) */

#define ARP_MODE_F16X3 3 /* arp_enc only: f32-accurate products on the 16-bit MFMA -- every GEMM operand is an (hi, lo) pair of binary16 values and a
                           product is hi.hi + lo.hi + hi.lo (three MFMAs, ~2^-22 relative; a K-concatenated operand [x_hi | x_lo | x_hi] against
                           [W_hi | W_hi | W_lo] on the ordinary f16 kernels), attention / LayerNorm / residual stream in f32.  The 16-bit mode of row N1
                           that meets north_star's 1e-3 on the policy logits (the plain f16 encoder does not: DESIGN 6b) */
#define ARP_MODE_F16C 4  /* arp_enc only (round 5): the binary16 encoder with the operand ROUNDINGS of its GEMMs corrected on the scaled fp4 MFMA --
                           a product is x_hi.W_hi (binary16) + 2^-s x4.dW4 (+ 2^-s' dx4.W4) where dW = W - W_hi, dx = x - x_hi are carried as e2m1
                           (OCP fp4) with power-of-two block scales: the correction terms are 2^-12 of the product, 1-2 significant bits take the
                           roundings out to a quarter of their size, and the fp4 MFMA moves four times the k per cycle on the same 4-register operand
                           tuples, so a product with both corrections costs 1.5x the binary16 one instead of f16x3's 3x.  Patch embedding on (hi, lo)
                           binary16 pairs; attention, LayerNorm statistics, residual stream as in ARP_MODE_F16.  ARP_F16C_PLAN selects, per GEMM
                           (in_proj, out_proj, fc1, fc2), 0 = plain / 1 = weight correction / 2 = both */

#define ARP_ACT_NONE 0
#define ARP_ACT_QGELU 1
#define ARP_ACT_RELU 2
#define ARP_ACT_TANH 3
#define ARP_ACT_GELU_TANH 4

const char* arp_last_error(void);
int arp_version(void);
int arp_device_count(void); /* number of visible HIP devices; 0 when there is none (never fails) */

/* ---- raw device memory + events (for HBM-resident benchmarking; no torch types anywhere) ------ */
int arp_dev_malloc(void** out, size_t bytes);
int arp_dev_free(void* p);
int arp_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int arp_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int arp_set_device(int device);
/* Pin / unpin a caller-owned host buffer (hipHostRegister): host-fed calls then move it by true asynchronous DMA.  For buffers that are
   reused across calls (registration costs milliseconds per 100 MB); unregister before freeing the memory. */
int arp_host_register(void* p, size_t bytes);
int arp_host_unregister(void* p);
int arp_dev_synchronize(void); /* hipDeviceSynchronize on the current device */

/* ---- path (1): CLIP reward labelling ------------------------------------------------------------
 * Replaces clip.load(...) + the compute_reward closure, arp_dt/label_reward.py:125-146
 * (third-party openai/CLIP forward; in-tree mirror arp_dt/models/openai/layers.py:274-449). */
typedef struct arp_clip arp_clip;

typedef struct arp_clip_cfg {
    int32_t patch;      /* 32 (BASELINE headline) or 16 (what the reference loads, label_reward.py:126) */
    int32_t width;      /* 768 */
    int32_t layers;     /* 12  */
    int32_t heads;      /* 12  */
    int32_t embed;      /* 512 */
    int32_t img_res;    /* 224 */
    int32_t txt_width;  /* 512 */
    int32_t txt_layers; /* 12  */
    int32_t txt_heads;  /* 8   */
    int32_t ctx;        /* 77  */
    int32_t vocab;      /* 49408 */
    int32_t mode;       /* ARP_MODE_F32 | ARP_MODE_BF16 | ARP_MODE_F16 */
    int32_t device;     /* HIP device ordinal */
    int32_t max_batch;  /* frames per internal pass (workspace size); <= 0 -> 1024 */
    int32_t attn_impl;  /* 0 = auto (MFMA kernel where available), 1 = force the VALU kernel */
    int32_t n_streams;  /* N in 2..4 = label a batch in N contiguous parts (each >= 128 frames) on N HIP streams so one
                           part's memory-bound kernels and GEMM tails overlap another part's GEMMs; 0/1 = one stream */
} arp_clip_cfg;

int arp_clip_create(const arp_clip_cfg* cfg, arp_clip** out);
int arp_clip_destroy(arp_clip* h);

/* One tensor of an openai/CLIP state dict (names and layouts: arp_dt/models/openai/model.py:220-314,
 * SURVEY.md Appendix B), f32 host data, row-major.  Call arp_clip_finalize_weights() after the last. */
int arp_clip_load_weight(arp_clip* h, const char* name, const float* data, const int64_t* shape, int ndim);
/* Before arp_clip_finalize_weights: run the vision tower's c_fc / c_proj GEMMs on the scaled fp8 MFMA (e4m3 operands, f32 accumulate;
 * BASELINE.json configs[4] "bf16 with fp8 MFMA GEMMs").  A THROUGHPUT mode for the frozen towers of the fine-tune step
 * (finetune_module/clip_multiscale_adapter.py:134-175): 3 significand bits, features ~1e-2 off the f32 towers -- not for labelling
 * to 1e-4.  16-bit modes only, width % 128 == 0.  on = 2: the attention's in_proj / out_proj on fp8 operands as well, in blocks whose QKV +
 * attention do not run on the fused kernel (ViT-B/16's 197 tokens): ln_1 and the attention output are written as e4m3. */
int arp_clip_set_fp8_mlp(arp_clip* h, int on);
int arp_clip_finalize_weights(arp_clip* h);

/* clip.tokenize output [n_prompts, ctx] int32 -> runs the text tower ONCE and caches the normalised
 * text features (the reference re-runs it per trajectory, label_reward.py:136-141). */
int arp_clip_set_text(arp_clip* h, const int32_t* tokens, int n_prompts);
int arp_clip_get_text_features(arp_clip* h, float* out /* [n_prompts, embed], L2-normalised */);
/* Which reduction over the cached prompts a reward is: 0 (default) = logits_per_text[0], prompt 0 whatever was cached (the offline pass,
 * arp_dt/label_reward.py:146); 1 = logits_per_text.mean(axis=0) over all cached prompts -- the rollout loop's live branch for a LIST of prompts
 * (arp_dt/envs/vl_reward.py:19-22).  Applies to every labelling call of the handle until changed. */
int arp_clip_set_prompt_reduce(arp_clip* h, int mode);

/* compute_reward (label_reward.py:132-146): uint8 NHWC frames -> float32 rewards
 * = exp(logit_scale) * cos(image, prompt 0).  use_crop selects the transform of label_reward.py:92-102. */
int arp_clip_label(arp_clip* h, const uint8_t* frames_host, int n, int H, int W, int use_crop, float* rewards_host);
/* Same, frames and rewards already in device memory; enqueued on the handle's stream. */
/* Asynchronous form of arp_clip_label for a STREAM of calls (n <= max_batch each): submit() enqueues upload + pass + download on slot
 * 0 / 1 and returns, collect() waits for that slot.  Submitting call i+1 before collecting call i keeps the compute streams full across
 * calls (a synchronous call pays the pipeline's fill and drain every time).  `frames` must stay valid until collect() of the slot. */
int arp_clip_label_submit(arp_clip* h, int slot, const uint8_t* frames_host, int n, int H, int W, int use_crop);
int arp_clip_label_collect(arp_clip* h, int slot, float* rewards_host);
int arp_clip_label_dev_async(arp_clip* h, const uint8_t* frames_dev, int n, int H, int W, int use_crop,
                             float* rewards_dev);
int arp_clip_sync(arp_clip* h);
int arp_clip_set_streams(arp_clip* h, int n_streams); /* change the stream count, 0..4 (see arp_clip_cfg.n_streams) */

/* model.encode_image (label_reward.py:156, clip_goal_conditioned variant) */
int arp_clip_encode_image(arp_clip* h, const uint8_t* frames_host, int n, int H, int W, int use_crop, int normalize,
                          float* out /* [n, embed] */);

/* The torchvision/PIL transform alone (label_reward.py:109-121 / :92-102), for PIL-parity tests:
 * uint8 NHWC -> f32 NCHW [n,3,res,res]. */
int arp_preprocess(const uint8_t* frames_host, int n, int H, int W, int use_crop, int res, float* out_nchw_host);
/* Host-side resample table (Pillow precompute_coeffs + normalize_coeffs_8bpc); no GPU needed.
 * xmin[out], cnt[out], weights[out*ksize_cap]; returns the max tap count or < 0. */
int arp_bicubic_coeffs(int in_size, int out_size, int32_t* xmin, int32_t* cnt, int32_t* weights, int ksize_cap);

/* Per-call-site HIP-event profile of the kernels launched on the handle's stream. */
int arp_clip_profile_enable(arp_clip* h, int on);
int arp_clip_profile_reset(arp_clip* h);
/* JSON object {"site": {"ms": total_ms, "calls": n}, ...}; returns bytes written or < 0. */
int arp_clip_profile_json(arp_clip* h, char* buf, int buf_len);

/* The shader clock the chip holds under the labelling pass (bench.py: whole_pass.clock_ghz).  on = 1 zeroes the accumulators and routes the vision
 * tower's c_fc launches through a DIAGNOSTIC instance of the same GEMM kernel that stamps s_memtime / s_memrealtime around every workgroup (the product
 * instances execute no stamp); rewards are unchanged.  arp_clip_clock_read: out[0] = GHz (sum of shader cycles / sum of 100 MHz ticks over the probed
 * workgroups), out[1] = workgroups probed, out[2] = mean workgroup duration in microseconds. */
int arp_clip_clock_probe(arp_clip* h, int on);
int arp_clip_clock_read(arp_clip* h, double* out3);

/* HIP events on the handle's stream (bench timing) */
typedef struct arp_event arp_event;
int arp_event_create(arp_event** out);
int arp_event_destroy(arp_event* e);
int arp_clip_event_record(arp_clip* h, arp_event* e);
int arp_event_elapsed_ms(arp_event* start, arp_event* stop, float* ms); /* synchronises on stop */

/* ---- path (2): ARP-DT policy train step ------------------------------------------------------------
 * Replaces create_train_step(...) -> train_step_fn(state, batch, rng), arp_dt/main_procgen.py:104-141
 * (model arp_dt/ARPDT.py:152-261,413-486 + arp_dt/layers.py; optimizer main_procgen.py:490-507).
 * Boundary: the frozen (stop_gradient) M3AE encodings are an input; adapter, image_text_input, token
 * embeddings, the causal transformer, both heads, losses, L2 term, gradient all-reduce, global-norm clip
 * and Adam are inside.  Tensors cross the boundary under their Flax tree path flattened with '/'
 * (e.g. "policy/Block_0/Attention_0/Dense_0/kernel"), in the Flax layout ([in, out] kernels). */
typedef struct arp_dt arp_dt;

typedef struct arp_dt_cfg {
    int32_t emb;         /* 128 */
    int32_t depth;       /* 2   */
    int32_t heads;       /* 8   */
    int32_t mlp_ratio;   /* 4   */
    int32_t n_actions;   /* 15  */
    int32_t window;      /* 4 time steps -> 12 tokens */
    int32_t enc_tokens;  /* 257 (M3AE ViT-B/16 at 256x256) */
    int32_t enc_dim;     /* 768 */
    int32_t use_adapter; /* 1   */
    int32_t mode;        /* ARP_MODE_F32 | ARP_MODE_F16 (default of arp_amd.train) | ARP_MODE_BF16: operand type of the adapter / image_text_input GEMMs */
    int32_t device;
    int32_t world;       /* data-parallel degree (set again by arp_dt_comm_init) */
    int32_t rank;
    float lambda_ret;    /* lambda_return_pred */
    float weight_decay;  /* coefficient of the explicit 0.5*wd*||p||^2 term (main_procgen.py:114-117) */
    float clip_norm;     /* optax.clip_by_global_norm */
    float b1, b2, eps;   /* adam */
    int32_t alibi_bias;  /* config.alibi_bias (arp_dt/layers.py:74-78; off in the shipped configuration): slope_h * key_index added to the policy
                            transformer's attention scores, slopes as _get_attention_slopes (layers.py:97-110) */
} arp_dt_cfg;

int arp_dt_create(const arp_dt_cfg* cfg, arp_dt** out);
int arp_dt_destroy(arp_dt* h);
int arp_dt_num_params(arp_dt* h, int64_t* total, int32_t* n_tensors);
int arp_dt_param_info(arp_dt* h, int i, char* name_buf, int name_len, int64_t* shape4, int32_t* ndim);
/* which: 0 = params, 1 = gradients (of the last backward), 2 = adam mu, 3 = adam nu */
int arp_dt_set_tensor(arp_dt* h, const char* name, int which, const float* data);
int arp_dt_get_tensor(arp_dt* h, const char* name, int which, float* out); /* which = 1 after a data-parallel step: the rank MEAN */
int arp_dt_set_step(arp_dt* h, int64_t step);
int arp_dt_get_step(arp_dt* h, int64_t* step);
/* Stages one per-device batch in HBM: enc f32 [B,T,enc_tokens,enc_dim], action int32 [B,T], rtg f32 [B,T,1]. */
int arp_dt_set_batch(arp_dt* h, const float* enc, const int32_t* action, const float* rtg, int B);
/* ARPDT.__call__ on the staged batch: logits [B,T,n_actions], return_pred [B,T,1],
 * metrics[4] = loss, acc (fraction), trans_loss, return_loss (any pointer may be NULL). */
int arp_dt_forward(arp_dt* h, float* action_logits, float* return_pred, float* metrics);
/* forward + backward + L2 term; gradients readable with arp_dt_get_tensor(.., 1, ..) (parity tests) */
int arp_dt_backward(arp_dt* h);
/* train_step_fn on the staged batch: forward, backward, L2 term, RCCL all-reduce (when a communicator is
 * attached), clip_by_global_norm, adam.  aux[9] = loss, acc*100, trans_loss, return_loss, weight_penalty,
 * weight_l2, train_state_step, learning_rate, (extra) gradient norm. */
int arp_dt_train_step(arp_dt* h, float lr, float* aux);
int arp_dt_train_step_async(arp_dt* h, float lr); /* no read-back; pair with arp_dt_sync */
/* val_step_fn of create_val_step (main_procgen.py:144-169): forward on the staged batch, rank mean (pmean, :165) of
   aux4 = {loss, trans_loss, return_loss, acc * 100}.  No parameter changes. */
int arp_dt_val_step(arp_dt* h, float* aux4);
/* Two device-resident batch slots (0 / 1) = prefetch_to_device(..., 2) of main_procgen.py:703.  upload_*_async copies a host batch
   into a slot on the handle's copy stream and returns without waiting for the GPU; it touches only that slot, so a prefetch thread
   may call it while arp_dt_train_step runs on the other slot (the copy itself waits, on the GPU, for the last step that read the
   slot).  select_batch makes the next forward / val_step / train_step read a slot, ordered behind its upload.  arp_dt_set_batch
   (synchronous) keeps writing the selected slot.  Pinned host memory makes the copy truly asynchronous; pageable works (staged). */
int arp_dt_upload_batch_async(arp_dt* h, int slot, const float* enc, const int32_t* action, const float* rtg, int B);
int arp_dt_upload_batch_images_async(arp_dt* h, int slot, const float* images, const int32_t* action, const float* rtg, int B);
int arp_dt_select_batch(arp_dt* h, int slot);
/* The data-parallel step all-reduces the flat gradient in two buckets of two ranges each; bucket 1 (image_text_input's kernel and
   everything the transformer owns) is launched on a communication stream while the adapter's backward GEMMs still run.
   ranges8 = {lo, hi} x 4 in floats (bucket 1: ranges 0, 1; bucket 2: ranges 2, 3), *total = the flat parameter count.  Needs no GPU.
   Environment: ARP_DT_OVERLAP=0 selects the serial form (one all-reduce behind the whole backward), ARP_DT_FORCE_COMM=1 runs the
   all-reduce path at world = 1 as well. */
int arp_dt_bucket_plan(const arp_dt_cfg* cfg, int64_t* ranges8, int64_t* total);
int arp_dt_sync(arp_dt* h);
int arp_dt_event_record(arp_dt* h, arp_event* e);
/* Data parallelism: one process per GPU.  Rank 0 calls arp_dt_comm_unique_id, ships the 128 bytes to the
 * others (any side channel), every rank calls arp_dt_comm_init, then arp_dt_broadcast_state
 * (= sync_state_fn, main_procgen.py:94-101). */
int arp_dt_comm_unique_id(void* id128);
int arp_dt_comm_init(arp_dt* h, const void* id128, int world, int rank);
int arp_dt_broadcast_state(arp_dt* h);
/* ARP_MODE_F16 only: the adapter's two FORWARD products (models/adapter/layers.py:19-30) with their binary16 operand roundings corrected on the scaled fp4 MFMA
 * (ARP_MODE_F16C's product) and the adapter output handed to the residual mix in f32: takes the policy's own share out of the encoder-inside logit error
 * (row N1) for ~0.1 ms per step.  Off by default (ARP_DT_ADAPTER_C=1 turns it on at create); the backward is unchanged. */
int arp_dt_set_adapter_corrections(arp_dt* h, int on);
/* test hook: the bytes of a named intermediate device buffer of the last forward ("Xc", "H1c", "W1c", "W2c", "wc_scal", "A32", "A", "Adx", "Y", "img", "H1", "Xb");
   copies min(bytes, size), returns the buffer's size or -1 */
int64_t arp_dt_debug_read(arp_dt* h, const char* name, void* out, int64_t bytes);
/* What the communicator says about itself, for a multi-GPU run that certifies itself (the pmean of main_procgen.py:132 needs every
 * device in it): info5 = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion code, 1 if a communicator exists};
 * without one (world 1) {1, 0, device, 0, 0}.  arp_dt_comm_selfcheck all-reduces (sum) the scalar rank + 1 through that
 * communicator on the step's stream: every rank must read world (world + 1) / 2 (without a communicator: 1, nothing reduced). */
int arp_dt_comm_info(arp_dt* h, int32_t* info5);
int arp_dt_comm_selfcheck(arp_dt* h, double* sum);
int arp_dt_profile_enable(arp_dt* h, int on);
int arp_dt_profile_reset(arp_dt* h);
int arp_dt_profile_json(arp_dt* h, char* buf, int buf_len);

/* ---- row N2: CLIP multi-scale adapter fine-tune step (BASELINE.json configs[4]) --------------------------------
 * The trainable head of finetune_module/clip_multiscale_adapter.py::CLIPMultiscaleAdapter on top of the FROZEN CLIP
 * towers (finetune_module/finetune.py:139-140): encode_image :134-149, encode_text :151-175, forward :177-250
 * (VIP loss + lambda_id * inverse-dynamics CE), optimiser torch.optim.AdamW over every non-CLIP parameter (:141).
 * The towers' outputs are the step's inputs: the per-block CLS (image) / EOT (text) features the reference collects with
 * forward hooks (finetune_module/utils.py:6-18), concatenated block 0..layers-1, and the final CLIP features.
 * Parameters cross under torch's state_dict names with torch's [out, in] Linear layout:
 *   "image_intermediate_linear.weight", "text_intermediate_linear.weight", "{image,text}_adapter.layers.{0,3}.{weight,bias}",
 *   "inverse_layer.layers.{0,3}.{weight,bias}", "image_residual_weight", "text_residual_weight", "lambda_id". */
typedef struct arp_ft arp_ft;
typedef struct arp_ft_cfg {
    int32_t layers;     /* clip_model.transformer.layers (12): blocks whose features are concatenated, both towers */
    int32_t width_v;    /* 768 */
    int32_t width_t;    /* 512 (= input_dim = output_dim of the reference constructor) */
    int32_t embed;      /* 512 */
    int32_t hidden;     /* hidden_dim 1024: adapters use hidden*(layers+1), the inverse model uses hidden */
    int32_t n_actions;  /* 15 */
    int32_t mode;       /* ARP_MODE_F32 | ARP_MODE_F16 (default of arp_amd.finetune) | ARP_MODE_BF16 */
    int32_t device;
    int32_t use_vip;    /* use_vip_loss */
    int32_t use_id;     /* use_id_loss */
    float gamma;        /* 0.98 (:107) */
    float logit_scale;  /* ln of the similarity scale, clip_model.logit_scale (:95) */
    float weight_decay; /* AdamW decoupled decay, finetune.py:31 (0.001) */
    float b1, b2, eps;  /* 0.9, 0.999, 1e-8 */
    int32_t goal_conditioned; /* clip_multiscale_adapter.py:208-212,224-230: the batch carries FOUR image groups (image0..image3); image3's
                                 adapted feature takes the prompt's place -- scores = -||a3 - a_k||, inverse-model input [a1|a3|a2|a3] --
                                 and the text head is never run (its parameters get no gradient and AdamW leaves them alone) */
} arp_ft_cfg;
/* The frozen towers' side of the step, on an arp_clip handle (ViT-B/16 in the reference, :118): per-block CLS / EOT
 * features and the un-normalised CLIP features.  Image frames go through the fine-tune transform of :120-132 (float,
 * bilinear resize to 224 when both sides differ, /255, normalise) -- NOT the PIL-bicubic one of label_reward.py; the
 * random ColorJitter of training (:24-36) belongs to the data pipeline and is not applied. */
int arp_clip_encode_image_multiscale(arp_clip* h, const uint8_t* frames_nhwc, int n, int H, int W,
                                     float* inter /* [n, layers*width] */, float* final_feat /* [n, embed] */);
int arp_clip_encode_text_multiscale(arp_clip* h, const int32_t* tokens /* [n, ctx] */, int n,
                                    float* inter /* [n, txt_layers*txt_width] */, float* final_feat /* [n, embed] */);
/* The same two calls with their OUTPUTS in device memory (arp_dev_malloc), for arp_ft_set_batch_dev: the tower features
 * then never cross PCIe. */
int arp_clip_encode_image_multiscale_dev(arp_clip* h, const uint8_t* frames_nhwc, int n, int H, int W, float* inter_dev, float* final_dev);
/* The same two outputs through the LABEL transform (Pillow-exact bicubic resize + normalise, use_crop as in arp_clip_label) instead of the
 * fine-tune transform: what the rollout loop's adapter rewards hand the fine-tuned model -- `model.encode_image(preprocess(Image.fromarray(obs)))`,
 * arp_dt/envs/vl_reward.py:44-61 (get_torch_clip_adapter_reward) and :64-79 (..._goal_conditioned_reward).  A call of a few frames runs on the
 * latency path. */
int arp_clip_encode_image_multiscale_pil(arp_clip* h, const uint8_t* frames, int n, int H, int W, int use_crop, float* inter, float* final_feat);
int arp_clip_encode_text_multiscale_dev(arp_clip* h, const int32_t* tokens, int n, float* inter_dev, float* final_dev);
int arp_ft_create(const arp_ft_cfg* cfg, arp_ft** out);
int arp_ft_destroy(arp_ft* h);
int arp_ft_num_params(arp_ft* h, int64_t* total, int32_t* n_tensors);
int arp_ft_param_info(arp_ft* h, int i, char* name_buf, int name_len, int64_t* shape4, int32_t* ndim);
/* which: 0 = parameter, 1 = gradient, 2 = AdamW exp_avg, 3 = AdamW exp_avg_sq.  Gradients: arp_ft_backward stores every one; a single-process
 * arp_ft_train_step applies the seven big weight gradients inside their GEMMs and never stores them -- reading one of those afterwards is an error
 * (ARP_FT_FUSE_ADAM=0 restores the stored gradients and the separate AdamW pass). */
int arp_ft_set_tensor(arp_ft* h, const char* name, int which, const float* data);
int arp_ft_get_tensor(arp_ft* h, const char* name, int which, float* out);
/* f16 mode only: gradient elements that arrived at AdamW as inf / NaN (a binary16 overflow in the x 1024-seeded backward) and were treated
   as missing for that step -- cumulative since create.  The other modes never mask: a non-finite gradient shows as NaN parameters, as in
   torch.optim.AdamW (finetune_module/finetune.py:141). */
int arp_ft_dropped_gradients(arp_ft* h, uint64_t* count);
int arp_ft_set_step(arp_ft* h, int64_t step);
int arp_ft_get_step(arp_ft* h, int64_t* step);
/* img_inter [3, B, layers*width_v] and img_final [3, B, embed]: frames image0, image1, image2 of each sample
 * (clip_multiscale_adapter.py:185-202; image3 is only read when goal_conditioned); txt_inter [B, layers*width_t],
 * txt_final [B, embed]; r [B] as stored in the batch (the loss uses r - 1, :215); action [B] class ids. */
int arp_ft_set_batch(arp_ft* h, const float* img_inter, const float* img_final, const float* txt_inter, const float* txt_final,
                     const float* r, const int32_t* action, int B);
int arp_ft_set_batch_dev(arp_ft* h, const float* img_inter_dev, const float* img_final_dev, const float* txt_inter_dev,
                         const float* txt_final_dev, const float* r /* host */, const int32_t* action /* host */, int B);
/* metrics4: loss, vip_loss, id_loss, lambda_id.  scores [3, B] and logits [B, n_actions] may be NULL. */
int arp_ft_forward(arp_ft* h, float* metrics4, float* scores, float* logits);
/* Inference half of one tower's head -- model.encode_image (which = 0) / model.encode_text (which = 1) as the clip_ft
 * labelling branch calls them (arp_dt/label_reward.py:165-230): tower features [n, layers*width] + [n, embed] -> adapted,
 * L2-normalised features [n, layers*width_t + embed].  Drops a batch staged with arp_ft_set_batch. */
int arp_ft_encode(arp_ft* h, int which, const float* inter, const float* final_feat, int n, float* out);
int arp_ft_backward(arp_ft* h);                        /* forward + backward: fills every gradient */
int arp_ft_train_step(arp_ft* h, float lr, float* aux4); /* forward + backward + AdamW; aux4 as metrics4 (pre-update) */
int arp_ft_train_step_async(arp_ft* h, float lr);
int arp_ft_sync(arp_ft* h);
int arp_ft_event_record(arp_ft* h, arp_event* e);
/* Data parallelism for the head step (BASELINE.json configs[4]: DP = 8; the reference's finetune.py is single-GPU): one process
 * per GPU, id from arp_dt_comm_unique_id on rank 0, then every step all-reduces the flat f32 gradient once (sum; the mean is
 * folded into the AdamW kernel) -- the scheme of the policy step (main_procgen.py:128-139). */
int arp_ft_comm_init(arp_ft* h, const void* id128, int world, int rank);
int arp_ft_broadcast_state(arp_ft* h);
int arp_ft_comm_info(arp_ft* h, int32_t* info5);       /* as arp_dt_comm_info / arp_dt_comm_selfcheck, on the head step's communicator */
int arp_ft_comm_selfcheck(arp_ft* h, double* sum);
/* The data-parallel step all-reduces the 1.9 GB gradient in seven buckets in the order the backward produces them (inverse model;
   per tower: second adapter layer, first adapter layer, intermediate linear), each launched on a communication stream as soon as its
   last weight-gradient GEMM is enqueued.  ranges28 = {lo, hi} x 14 in floats (bucket b = ranges 2b, 2b + 1; empty ranges have
   lo = hi), *total = the flat parameter count.  Needs no GPU.  ARP_FT_OVERLAP=0: one all-reduce behind the whole backward;
   ARP_FT_FORCE_COMM=1: the all-reduce path at world = 1 too.  arp_ft_get_tensor(which = 1) after such a step: the rank MEAN. */
int arp_ft_bucket_plan(const arp_ft_cfg* cfg, int64_t* ranges28, int64_t* total);
int arp_ft_profile_enable(arp_ft* h, int on);
int arp_ft_profile_reset(arp_ft* h);
int arp_ft_profile_json(arp_ft* h, char* buf, int buf_len);

/* ---- row N1: frozen M3AE image encoder (forward_representation) -----------------------------------
 * arp_dt/models/m3ae/model.py:471-496 as called under stop_gradient by arp_dt/ARPDT.py:413-462.
 * Weights cross under their Flax tree path ('/'-flattened, [in,out] kernels): "cls_token",
 * "encoder_image_type_embedding", "image_embedding/{kernel,bias}", "encoder/Block_i/...", "encoder/LayerNorm_0/...". */
typedef struct arp_enc arp_enc;
typedef struct arp_enc_cfg {
    int32_t patch;      /* 16  */
    int32_t width;      /* 768 */
    int32_t layers;     /* 12  */
    int32_t heads;      /* 12  */
    int32_t mlp_ratio;  /* 4   */
    int32_t img_res;    /* 256 -> 257 tokens */
    int32_t mode;       /* ARP_MODE_F32 | ARP_MODE_F16 | ARP_MODE_BF16 | ARP_MODE_F16X3 | ARP_MODE_F16C (need not equal the mode of the policy handle it is attached to) */
    int32_t device;
    int32_t max_frames; /* frames per internal pass; <= 0 -> 128 */
    int32_t attn_impl;  /* 0 auto, 1 VALU */
} arp_enc_cfg;
int arp_enc_create(const arp_enc_cfg* cfg, arp_enc** out);
int arp_enc_destroy(arp_enc* h);
int arp_enc_load_weight(arp_enc* h, const char* name, const float* data, const int64_t* shape, int ndim);
int arp_enc_finalize_weights(arp_enc* h);
/* images f32 NHWC [n,res,res,3] (already normalised, as the training pipeline delivers them) -> f32 [n,tokens,width] */
int arp_enc_forward(arp_enc* h, const float* images_host, int n, float* out_host);
/* Part streams (round 6): a call's frames are encoded as `n_streams` contiguous parts (1..4, default 2), part 0 on the caller's stream and the others on
 * streams of their own, so that one part's memory-bound kernels and GEMM grid tails run beside another part's GEMMs -- as arp_clip_cfg.n_streams does
 * for the labelling pass.  first_part_frames > 0: that many frames in part 0 of two; 0: the default cut (15/32 of the frames: parts that end at different times
 * fill each other's grid tails better, -1.5 % per step); < 0: equal parts.  A part of fewer than min_part_frames frames (<= 0: keep the default, 24) is not cut off.  Encodings are bit-identical for every setting. */
int arp_enc_set_streams(arp_enc* h, int n_streams, int first_part_frames, int min_part_frames);
int arp_enc_profile_enable(arp_enc* h, int on);
int arp_enc_profile_json(arp_enc* h, char* buf, int buf_len);
/* Put the frozen encoder INSIDE the policy step: after attaching, arp_dt_set_batch_images stages raw frames
 * [B,T,res,res,3] f32 and every forward / train step runs the encoder first (on the step's stream). */
int arp_dt_attach_encoder(arp_dt* h, arp_enc* enc);
int arp_dt_set_batch_images(arp_dt* h, const float* images, const int32_t* action, const float* rtg, int B);
/* The encoder is frozen: batch i + 1's encodings depend on nothing step i computes.  arp_dt_encode_ahead(slot) enqueues the encoder pass of the frames in
 * batch slot `slot` (0 / 1: uploaded with arp_dt_upload_batch_images_async; 2: staged by arp_dt_set_batch_images) on the encoder's own stream NOW -- behind the
 * slot's upload and the last step that read the slot -- so that it runs beside the current step's policy part; the step that then reads the slot waits for it
 * instead of encoding.  Callable from the uploader thread.  Without it a step encodes its own batch at its head (same stream of work, same encodings). */
int arp_dt_encode_ahead(arp_dt* h, int slot);

/* ---- host I/O of path (1) (no GPU involved) -------------------------------
 * Reads n stored chunks of a gzip-chunked dataset (`ob` of data/PPG/trajectory_recorder.py:148-176: one chunk = one row =
 * num_frames frames) from file descriptor fd at addr[i] (size[i] bytes as stored; 0 = never written), inflates them on
 * `threads` native threads (<= 0: one per hardware thread) and copies the LAST cnt[i] frames of chunk i to out + dst_off[i].
 * stored_raw[i] != 0 (may be NULL): the deflate filter was skipped for that chunk.  Replaces the inflate half of
 * g[img_key][traj, -1] (arp_dt/label_reward.py:268); chunk addresses come from libhdf5 (arp_amd/h5store.py). */
int arp_h5_inflate_last_frames(int fd, int n, const uint64_t* addr, const uint64_t* size, const uint8_t* stored_raw,
                               uint64_t chunk_bytes, uint64_t frame_bytes, const uint64_t* dst_off, const uint32_t* cnt,
                               uint8_t* out, int threads);
/* Reward datasets of label_reward.py:273-289 (gzip chunks of ONE row): rows [first_row, first_row + n_rows) of a dataset whose chunk
   is (1, row...) are deflated here with one reused zlib stream and placed with H5Dwrite_chunk -- `write_chunk_fn` is that function's
   address inside the libhdf5 the caller has loaded (hid_t = int64, HDF5 >= 1.10.3).  `level` = the dataset's deflate level. */
int arp_h5_write_rows_deflated(void* write_chunk_fn, int64_t dset, int64_t dxpl, const void* rows, uint64_t row_bytes, uint64_t n_rows,
                               uint64_t first_row, int ndim, int level);

/* ---- single-operator entry points (host buffers; used by the per-kernel parity tests) ---------- */
/* out[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ resid), operands rounded to bf16 in ARP_MODE_BF16. */
/* fp8 (e4m3) instances of the 256x256 GEMM: operands rounded to e4m3 on the host; f32 output (+ residual) or e4m3 output
 * (out_scale * act(...), returned decoded); out_fp8 = 2: IEEE-half output (the in_proj of the fp8 attention projections, returned
 * decoded).  K % 128 == 0, N % 16 == 0. */
int arp_op_gemm_fp8(int act, const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N, int K, float alpha,
                    int out_fp8, float out_scale);
int arp_op_gemm_nt(int mode, int act, const float* A, const float* W, const float* bias, const float* resid,
                   float* out, int M, int N, int K);
/* The latency path's GEMM (csrc/skinny.h: W-tiled, M <= 1024 rows, N % 16, K % 32, 16-bit modes), the kernel behind the single-frame
 * reward (reference call site: arp_dt/envs/vl_reward.py:11-23).  ksplit = 0: one launch, out = act(A.W^T + bias) + resid;
 * ksplit >= 1 (K % (32 ksplit) == 0): split-K slabs + row kernel, out = resid + bias + A.W^T, and h_out = LayerNorm(out; ln_w, ln_b, eps)
 * rounded to the operand type when ln_w != NULL. */
int arp_op_skinny_gemm(int mode, int act, const float* A, const float* W, const float* bias, const float* resid, float* out, int M, int N, int K,
                       int ksplit, const float* ln_w, const float* ln_b, float eps, float* h_out);
/* Kernels of the policy train step (16-bit modes; reference math: arp_dt/ARPDT.py:462-484 and its autodiff).
 * arp_op_gemm_tn: C[M,N] = alpha * sum_k A[k,m] B[k,n], A [K,M] and B [K,N] row-major (weight gradients dW = dY^T X);
 *   tile256 = 0: 128x128 tiles (M, N % 128), 1: 256x256 tiles (M, N % 256); K % 64; ksplit >= 1 K-slices.
 * arp_op_gemm_relu_bwd: out = (A . W^T) * (mask > 0) rounded to the operand type, colsum[N] = column sums of out (N % 8, K % 64).
 * arp_op_adapter_dy: dApre[R, tokens*D] = sigmoid(rw) * (dz[R,E] . Wi[E, tokens*D]) * (A > 0), colsum[D] = sums of dApre over rows
 *   and tokens, dres[0] = sum (dz . Wi) * (A - x); E in {32, 64, 128}, D % 128 == 0. */
int arp_op_gemm_tn(int mode, int tile256, int ksplit, const float* A, const float* B, float* out, int M, int N, int K, float alpha);
int arp_op_gemm_relu_bwd(int mode, const float* A, const float* W, const float* mask, float* out, float* colsum, int M, int N, int K);
int arp_op_adapter_dy(int mode, const float* dz, const float* Wi, const float* A, const float* x, float rw, float* dApre, float* colsum,
                      float* dres, int R, int E, int tokens, int D);
/* Times `iters` launches of the GEMM on device-resident random operands (HIP events); kernel: 1 = 128x128,
 * 2 = 256x256 pipelined, 0 = auto.  act/resid/out_f32 select the epilogue.  Returns the average ms per launch. */
/* ARP_MODE_F16C's product alone (gemm256 MIXC): A [M,K], W [N,K] f32 -> operand rows [rn16 | e2m1 segments] built on the host exactly as the encoder's
   kernels build them, out = A.W^T + bias (f32) with the weight rounding (plan 1) or both operand roundings (plan 2) corrected on the scaled fp4 MFMA;
   plan 0 = the plain binary16 product.  sd_sw[2] (optional): the power-of-two exponents of the e2m1 weight segments.  K % 256, N % 8. */
int arp_op_gemm_f16c(int plan, const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int* sd_sw);
int arp_op_gemm_bench(int mode, int kernel, int act, int resid, int out_f32, int M, int N, int K, int iters, float* avg_ms);
int arp_op_layernorm(const float* x, const float* w, const float* b, float* out, int rows, int D, float eps);
/* qkv [B*N, 3*D] -> out [B*N, D]; impl 0 = MFMA (bf16 mode, head_dim 64 only), 1 = VALU. */
int arp_op_attention(int mode, int impl, const float* qkv, float* out, int B, int N, int D, int heads, int causal);

#ifdef __cplusplus
}
#endif
#endif /* ARP_HIP_H */
