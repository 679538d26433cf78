/*
 * s2v_hip.h -- C ABI of libs2v_hip.so: the MI355X (gfx950) implementation of the CogVideoX denoise hot path that
 * carpedkm/disentangled-subject-to-vid drives from src/custom_cogvideox_pipe.py.
 *
 * The reference has no FFI: its plug-in seams are Python protocols.  Each entry point below
 * names the reference interface it stands behind (paths relative to the reference checkout).  The ctypes binding a
 * maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every tensor argument is a DEVICE pointer to a dense row-major array of the model dtype (S2V_F32 / S2V_BF16)
 *     unless stated otherwise; the caller owns it and keeps it alive until the stream has passed the call;
 *   - weights are copied (and re-packed: fused QKV, stacked AdaLN linears, padded tiles) into context-owned buffers by
 *     s2v_load_weight, the caller may free its copy afterwards;
 *   - every compute entry point is asynchronous on the given hipStream_t;
 *   - return value 0 = ok, < 0 = error; s2v_last_error() returns a thread-local message; no C++ exception crosses;
 *   - one context must not be used from two host threads at once; distinct contexts (one per GPU) are independent.
 */
#ifndef S2V_HIP_H
#define S2V_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the shared library is built with -fvisibility=hidden: exactly the entry points declared here are exported */
#if defined(__GNUC__)
#define S2V_API __attribute__((visibility("default")))
#else
#define S2V_API
#endif

typedef struct s2v_ctx s2v_ctx;
typedef void* s2v_stream; /* hipStream_t */

/* F16 (round 5): the dtype the reference selects for every non-5B checkpoint (src/inference.py:191,209: transformer, VAE and text
 * encoder alike).  Transformer contexts (s2v_create), the VAE decoder / encoder (s2v_vae_create, s2v_vae_enc_create), the T5 encoder
 * (s2v_t5_create), the scheduler step and the operator-level entry points all take it. */
enum { S2V_DTYPE_F32 = 0, S2V_DTYPE_BF16 = 1, S2V_DTYPE_F16 = 2 };

/* CogVideoXTransformer3DModel.__init__ hyper-parameters
 * (diffusers/src/diffusers/models/transformers/cogvideox_transformer_3d.py:253-280) */
typedef struct s2v_model_config {
    int32_t num_layers;       /* 30 (2B) / 42 (5B) */
    int32_t num_heads;        /* 30 / 48; head_dim is fixed at 64 */
    int32_t in_channels;      /* 16 */
    int32_t out_channels;     /* 16 */
    int32_t patch_size;       /* 2 */
    int32_t time_embed_dim;   /* 512 */
    int32_t text_embed_dim;   /* 4096 */
    int32_t use_rope;         /* use_rotary_positional_embeddings: 0 (2B) / 1 (5B) */
    int32_t dtype;            /* S2V_DTYPE_* : storage + rounding points of the model */
    float norm_eps;           /* 1e-5 */
    int32_t force_simple;     /* 1 = run the model on the generic VALU kernels (bf16: instead of the bf16 MFMA path; fp32: instead of the fp32-MFMA kernels, which return the same GEMM bits); cross-check only */
    int32_t weight_format;    /* 0 = the model dtype; 1 = W8A8 fp8 (BASELINE configs[4]): the four big linears of every block
                               * (fused QKV, attention out, FF1, FF2) keep OCP e4m3 weights with per-output-channel scales
                               * (quantised by s2v_finalize_weights after any LoRA merge) and take per-token e4m3 activations,
                               * on v_mfma_scale_f32_32x32x64_f8f6f4; bf16 model dtype only, inner_dim % 128 == 0;
                               * 2 = 1 plus fp8 QK^T (an option BEYOND configs[4]'s "fp8 weights", off unless asked for): after the per-head LayerNorm and
                               * rotary embedding q (times scale * log2 e) and k are re-quantised as MX e4m3 (one power-of-two scale per
                               * 32 head-dim elements) and S = K.Q^T of the attention runs on the same scaled fp8 MFMA; softmax, P and P.V
                               * stay fp32 / bf16;
                               * 3 = 1 below 40 000 tokens per sample, 2 from there on (decided at s2v_set_geometry): at the configs[4] geometry
                               * (N = 50 626) the attention is > 80 % of the step, fp8 QK^T takes 11-14 % off it and adds nothing measurable to the
                               * fp8 engine's whole-run drift on synthetic weights (profiles/r05_whole_run_c5_10steps.txt).  Opt-in like 2: the package's
                               * configs[4] preset is 1; its "cogvideox-5b-fp8-auto" preset asks for 3.  s2v_fp8_qk_active reports the decision.
                               * The reference has no fp8 path: parity unpinned, tolerance stated in tests/test_gpu_fp8.py */
    int32_t lora_adaln_scope; /* where the subject-LoRA acts inside CogVideoXLayerNormZero (normalization.py:467-484):
                               * 0 = as shipped: `enable_lora([self.linear], False)` sets an attribute nothing reads, so the LoRA
                               *     is active on both evaluations of norm{1,2}.linear and is merged into it (SURVEY preamble 6);
                               * 1 = as the authors' comments intend (:470-476): base weights for the video / text modulation, the
                               *     LoRA only for the reference-image chunks (cond_shift, cond_scale, cond_gate): the context keeps
                               *     a second copy of rows [0, 3D) of each norm{1,2}.linear, s2v_merge_lora on those names merges
                               *     into that copy only, and the reference-image rows are modulated / gated with it */
    int32_t attn_p_format;    /* softmax probabilities P and V^T of the four-wave attention kernel (sequences > 4608 tokens, and the fp8 engines):
                               * 0 = bf16 (P.V on v_mfma_f32_32x32x16_bf16, row sums in fp32, deferred maximum 2^64);
                               * 1 = fp16 (P.V on v_mfma_f32_32x32x16_f16; P keeps 11 significant bits instead of 8; row sums by packed fp16
                               *     adds on the P registers, flushed to fp32 per KV tile; the deferred maximum falls to 2^14, i.e. the slow
                               *     path re-adopts the row maximum when a later score exceeds it by ~9.7 natural units -- faster on smooth
                               *     score distributions, slower on spiky ones; tests/test_gpu_parity.py holds both against fp64 SDPA) */
    int32_t reserved[2];
} s2v_model_config;

S2V_API const char* s2v_last_error(void);
S2V_API const char* s2v_version(void);

S2V_API int s2v_create(const s2v_model_config* cfg, s2v_ctx** out);
S2V_API void s2v_destroy(s2v_ctx* ctx);

/* Load one state-dict tensor.  `name` is the reference's own key, e.g.
 * "transformer_blocks.3.attn1.to_q.weight", "patch_embed.proj.weight" [D,16,2,2], "norm_out.linear.bias"
 * (key list: SURVEY.md section 8b).  src_dtype may differ from the model dtype (converted on the fly).
 * Replaces ModelMixin.from_pretrained/.to(device,dtype) for this model (src/inference.py:191-215). */
S2V_API int s2v_load_weight(s2v_ctx* ctx, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim,
                    int32_t src_dtype, s2v_stream stream);
/* W[name] += scale * B.A   (A:[r,in] B:[out,r], fp32 device tensors; Conv2d patch_embed.proj takes A as [r,in*k*k]).
 * The merge the reference's PEFT LoRA is equivalent to (src/inference.py:218-229, alpha/r = 0.5).
 * Must be called after s2v_load_weight(name) and before s2v_finalize_weights. */
S2V_API int s2v_merge_lora(s2v_ctx* ctx, const char* name, const float* A, const float* B, int32_t rank, float scale,
                   s2v_stream stream);
/* Checks that every tensor was loaded; after this call the weights are immutable. */
S2V_API int s2v_finalize_weights(s2v_ctx* ctx, s2v_stream stream);
/* Total bytes of context-owned weights (for the broadcast) and access to the packed arena so that ONE
 * collective can replicate a finalized model rank0 -> all. */
S2V_API int s2v_weight_arena(s2v_ctx* ctx, void** dev_ptr, int64_t* bytes);
/* Where one state-dict tensor lives inside the arena: byte offset of element [0][0], its rows x cols and the leading dimension
 * in elements of the model dtype (rows of the fused QKV / stacked modulation buffers are `ld` apart).  What a checkpoint tool
 * needs to read a merged weight back (W + (alpha/r) B A of src/inference.py:218-229) without knowing the packing. */
S2V_API int s2v_weight_slot(s2v_ctx* ctx, const char* name, int64_t* offset_bytes, int64_t* rows, int64_t* cols, int64_t* ld);

/* Token geometry of the next calls: batch B (2 = CFG pair), text tokens T, latent frames F and latent H x W
 * (R = (H/2)(W/2) reference-image tokens, V = F*R video tokens, sequence order [text | ref | video]).
 * Allocates the activation workspace (no allocation happens inside the compute calls). */
S2V_API int s2v_set_geometry(s2v_ctx* ctx, int32_t B, int32_t T, int32_t F, int32_t H, int32_t W);

/* whether QK^T of the attention runs in MX e4m3 at the current geometry: weight_format 2 always, 3 from its token threshold on (the decision
 * s2v_set_geometry took), 0 otherwise -- callers report this instead of re-deriving the threshold */
S2V_API int s2v_fp8_qk_active(s2v_ctx* ctx, int32_t* active);

/* RoPE tables, fp32 [R + V, 64] (reference rows first): cos/sin as produced by
 * get_3d_rotary_pos_embed (embeddings.py:505-570) and sliced in custom_cogvideox_pipe.py:223-235.
 * A setup call: it synchronises `stream` and inspects the tables on the host once per geometry -- tables that repeat every value
 * twice (repeat_interleave(2), what the reference builds) are also kept in a packed form that lets the QKV projection apply the
 * rotary embedding in its epilogue; any other table is honoured through the separate q/k pass. */
S2V_API int s2v_set_rope(s2v_ctx* ctx, const float* cos_dev, const float* sin_dev, s2v_stream stream);
/* 2B only: additive 3-D sincos table for the video tokens, model dtype [V, D] (embeddings.py:380-401,440-446). */
S2V_API int s2v_set_pos_embed(s2v_ctx* ctx, const void* table_dev, s2v_stream stream);

/* Step-invariant conditioning: text [B,T,text_embed_dim] -> patch_embed.text_proj; ref image latent [1,1,C,H,W] ->
 * patch_embed.proj, duplicated over the batch (cogvideox_transformer_3d.py:494-504). */
S2V_API int s2v_set_conditioning(s2v_ctx* ctx, const void* text_dev, const void* ref_latent_dev, s2v_stream stream);

/* CogVideoXTransformer3DModel.forward (cogvideox_transformer_3d.py:450-560) with eval=True.
 * latents [B,F,C,H,W] (lat_bstride = elements between samples, 0 = all samples share one latent), timesteps fp32
 * DEVICE [B]; out [B,F,C,H,W]. */
S2V_API int s2v_transformer_forward(s2v_ctx* ctx, const void* latents, int64_t lat_bstride, const float* timesteps_dev,
                            void* out, s2v_stream stream);

/* CogVideoXBlock.forward (cogvideox_transformer_3d.py:122-186) for layer `layer`: three residual streams in/out,
 * hidden [B,V,D], enc0 (text) [B,T,D], enc1 (ref) [B,R,D], temb [B,time_embed_dim]. */
S2V_API int s2v_block_forward(s2v_ctx* ctx, int32_t layer, const void* hidden, const void* enc0, const void* enc1,
                      const void* temb, void* out_hidden, void* out_enc0, void* out_enc1, s2v_stream stream);

/* CogVideoXAttnProcessor2_0.__call__ (attention_processor.py:2024-2097) with layer `layer`'s attn1 weights:
 * hidden [B,V,D] and encoder [B,T+R,D] are the already-modulated inputs; outputs have the same shapes. */
S2V_API int s2v_attn_forward(s2v_ctx* ctx, int32_t layer, const void* hidden, const void* encoder, void* out_hidden,
                     void* out_encoder, s2v_stream stream);

/* Per-step scheduler scalars, computed on the host exactly as scheduling_ddim_cogvideox.py:364-394 /
 * scheduling_dpm_cogvideox.py:306-434 do (fp64), then cast; see s2v schedulers.py. */
typedef struct s2v_sched_coef {
    int32_t kind;      /* 0 DDIM, 1 DPM first/last step, 2 DPM multistep */
    float guidance;    /* classifier-free guidance scale of this step */
    float c_x0_x, c_x0_v, a_t, b_t, m1, m2, m3, m4, mn;
    float pad;
} s2v_sched_coef;

/* scheduler.step (+ optional CFG combine, custom_cogvideox_pipe.py:266-296); ctx may be NULL.
 * flags bit0: noise_pred is the CFG pair [2,n] (uncond, cond) and is combined with coef->guidance;
 *       bit1: noise_pred is fp32 (what the reference hands to scheduler.step after .float());
 *       bit2: latents_out is fp32 and un-rounded (scheduler.step's own return value) instead of `dtype`.
 * latents in/out [n] (may alias); x0_hist fp32 [n] (DPM: read as old x0, then overwritten; may be NULL for DDIM);
 * noise [n] in `dtype` (DPM only). */
S2V_API int s2v_sched_step(s2v_ctx* ctx, const s2v_sched_coef* coef_host, const void* noise_pred, int32_t flags,
                   const void* latents_in, void* latents_out, float* x0_hist, const void* noise, int64_t n,
                   int32_t dtype, s2v_stream stream);

/* One iteration of the denoise loop (custom_cogvideox_pipe.py:241-296): transformer on the CFG pair sharing
 * `latents` [1,F,C,H,W], fp32 CFG, scheduler step, round to the model dtype; latents updated IN PLACE.
 * use_graph != 0 captures the launch sequence into a hipGraph on first use and replays it afterwards
 * (timestep and coefficients live in device memory, so one graph serves all steps). */
S2V_API int s2v_denoise_step(s2v_ctx* ctx, void* latents, float timestep, const s2v_sched_coef* coef_host, float* x0_hist,
                     const void* noise, int32_t use_graph, s2v_stream stream);
/* pointer to the [B,F,C,H,W] model output of the last s2v_denoise_step (context-owned, model dtype) */
S2V_API int s2v_last_noise_pred(s2v_ctx* ctx, void** dev_ptr);

/* Live per-kernel timing for the roofline report (bench.py): HIP events are recorded on the launch stream around
 * every launch of a class; classes 0 qkv GEMM, 1 attention, 2 out-proj GEMM, 3 FF1 GEMM, 4 FF2 GEMM,
 * 5 LN-modulate, 6 qk-norm/rope/V^T.  Not recorded inside a captured graph.  s2v_profile_read synchronises the
 * device, returns total ms and launch counts per class since the previous read, and resets the counters. */
S2V_API int s2v_profile_enable(s2v_ctx* ctx, int32_t on);
S2V_API int s2v_profile_read(s2v_ctx* ctx, float* ms_by_class, int32_t* launches_by_class, int32_t nclass);
/* Average SHADER CLOCK (MHz) under the profiled launches of every class since s2v_profile_enable(1): one designated workgroup of the
 * profiled kernel (the four-wave GEMM / attention kernels; workgroup 0 of a persistent launch, the middle one otherwise) reads s_memtime
 * (shader-clock cycles) and s_memrealtime (constant 100 MHz) at its entry and exit; clock = sum of cycle spans / sum of tick spans x 100;
 * 0 for a class whose kernels carry no stamps.  The part is power-managed (1.1-1.9 GHz under the MFMA loops against the 2.4 GHz
 * the datasheet peak assumes): this is the figure that relates a measured TFLOP/s to the matrix pipe's cycles.  Synchronises. */
S2V_API int s2v_profile_read_clocks(s2v_ctx* ctx, float* mhz_by_class, int32_t nclass);
/* Marks every tensor as loaded on a replica whose arena was filled by a broadcast of s2v_weight_arena. */
S2V_API int s2v_mark_weights_loaded(s2v_ctx* ctx);

/* ---- replicas over RCCL: one process per GPU, ONE collective -- the weight broadcast root -> all ----------
 * No reference code (the reference is single-process); replaces what a maintainer would otherwise write around
 * torch.distributed.broadcast.  RCCL is bound at first use (dlopen librccl.so.1; a process that already loaded PyTorch-ROCm's copy
 * shares it); every call fails with a message when it is absent.  Protocol: the root calls s2v_rccl_unique_id, ships the 128 bytes
 * to the other ranks by any means (a file, a socket, torch.distributed's store), every rank -- with ITS GPU current -- calls
 * s2v_rccl_comm_create, then s2v_bcast_weights / s2v_rccl_bcast on a stream of that GPU. */
typedef struct s2v_rccl_comm s2v_rccl_comm;
/* 0 when librccl and every symbol this file needs can be bound (dlopen + dlsym only: no bootstrap thread, no socket, no communicator);
 * < 0 with the reason in s2v_last_error() otherwise.  What every rank calls before the ranks agree on the native path. */
S2V_API int s2v_rccl_available(void);
S2V_API int s2v_rccl_unique_id(void* id128 /* out: 128 bytes */);
S2V_API int s2v_rccl_comm_create(const void* id128, int32_t rank, int32_t world, s2v_rccl_comm** out);
S2V_API void s2v_rccl_comm_destroy(s2v_rccl_comm* comm);
/* any device range (s2v_vae_weight_arena, s2v_t5_weight_arena ...), in place, asynchronous on `stream`, 256-MiB collectives */
S2V_API int s2v_rccl_bcast(s2v_rccl_comm* comm, void* dev_ptr, int64_t bytes, int32_t root, s2v_stream stream);
/* ncclAllGather of bytes: every rank contributes bytes_per_rank from `send`; `recv` (world x bytes_per_rank, rank order) is the same on every
 * rank afterwards; send == recv + rank * bytes_per_rank is the in-place form.  The per-step exchange of CFG-parallel below. */
S2V_API int s2v_rccl_allgather(s2v_rccl_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, s2v_stream stream);
/* the transformer's finalized arena (merged LoRA, fused QKV, fp8 copies + scales) root -> all; receivers are marked loaded */
S2V_API int s2v_bcast_weights(s2v_ctx* ctx, s2v_rccl_comm* comm, int32_t root, s2v_stream stream);

/* ---- CFG-parallel: ONE video on TWO GPUs (SURVEY.md section 8e "noted for later"; round 6) -----------------------------------------
 * The CFG pair the reference batches (custom_cogvideox_pipe.py:255-265: latents duplicated, [negative | positive] embeddings, reference tokens
 * duplicated x2 at cogvideox_transformer_3d.py:503-504) is two independent forwards that meet only in the guidance formula (:266-279).  Each rank
 * of a pair sets a B = 1 geometry, s2v_set_conditioning with ITS half of the embeddings ([1,T,text_embed_dim]) and the one reference latent, and
 * runs per step:
 *   s2v_denoise_split_begin  the rank's forward into half `slot` (0 = unconditional / negative prompt, 1 = conditional) of the context's pair
 *                            buffer; use_graph != 0 replays a captured forward-only hipGraph;
 *   the exchange             s2v_rccl_allgather on s2v_cfg_pair's buffer in place (or any other transport), OUTSIDE the graph;
 *   s2v_denoise_split_end    fp32 CFG + scheduler step + round to the model dtype (:266-296) on the pair, by BOTH ranks redundantly: their latents
 *                            stay bit-identical without a second collective (DPM: the caller hands both ranks the same `noise`).
 * s2v_denoise_step_cfg_parallel is the three in one call over the library's own communicator (a 2-rank s2v_rccl_comm).  Per sample the arithmetic
 * is the B = 2 engine's: tests/test_gpu_cfg_parallel.py holds the pair to s2v_denoise_step bit for bit. */
S2V_API int s2v_denoise_split_begin(s2v_ctx* ctx, const void* latents, float timestep, const s2v_sched_coef* coef_host, int32_t slot,
                                    int32_t use_graph, s2v_stream stream);
/* the pair buffer [2][F,C,H,W] (model dtype, context-owned) and the bytes of one half */
S2V_API int s2v_cfg_pair(s2v_ctx* ctx, void** dev_ptr, int64_t* bytes_per_half);
/* one end per begin (the step's coefficients are the ones begin uploaded); x0_hist / noise as s2v_denoise_step (required for the DPM kinds) */
S2V_API int s2v_denoise_split_end(s2v_ctx* ctx, void* latents, float* x0_hist, const void* noise, s2v_stream stream);
/* comm: a communicator of exactly the pair's two ranks whose rank equals `slot` (the in-place all-gather puts rank r's bytes into half r) */
S2V_API int s2v_denoise_step_cfg_parallel(s2v_ctx* ctx, s2v_rccl_comm* comm, int32_t slot, void* latents, float timestep,
                                          const s2v_sched_coef* coef_host, float* x0_hist, const void* noise, int32_t use_graph,
                                          s2v_stream stream);

/* ---- CogVideoX 3-D causal VAE decode ------------------------------------------------------------------------- */
typedef struct s2v_vae s2v_vae;
/* AutoencoderKLCogVideoX.__init__ (models/autoencoders/autoencoder_kl_cogvideox.py:1020-1052), decoder part */
typedef struct s2v_vae_config {
    int32_t latent_channels;          /* 16 */
    int32_t out_channels;             /* 3 */
    int32_t num_blocks;               /* len(block_out_channels) = 4 */
    int32_t block_out_channels[8];    /* (128, 256, 256, 512): encoder order, the decoder walks it reversed */
    int32_t layers_per_block;         /* 3 */
    int32_t norm_num_groups;          /* 32 */
    int32_t temporal_compression_ratio; /* 4 */
    int32_t sample_height, sample_width; /* 480, 720: only used for the tile geometry */
    int32_t dtype;                    /* S2V_DTYPE_* */
    int32_t force_simple;             /* 1 = generic kernels only (cross-check) */
    float scaling_factor;             /* 1.15258426 (2B) / 0.7 (5B) */
    float norm_eps;                   /* 1e-6 */
    int32_t reserved[4];
} s2v_vae_config;

S2V_API int s2v_vae_create(const s2v_vae_config* cfg, s2v_vae** out);
S2V_API void s2v_vae_destroy(s2v_vae* vae);
/* name = the reference's state-dict key ("decoder.conv_in.conv.weight", "decoder.up_blocks.1.resnets.0.conv_shortcut.
 * weight", "decoder.mid_block.resnets.0.norm1.conv_y.conv.bias", ...); conv weights are re-packed to
 * [cout][(dt,dy,dx)][cin] for the channels-last implicit GEMM. */
S2V_API int s2v_vae_load_weight(s2v_vae* vae, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim,
                        int32_t src_dtype, s2v_stream stream);
S2V_API int s2v_vae_finalize(s2v_vae* vae);
/* Replicas ("11.14 GB transformer + 0.25 GB VAE decoder" are broadcast rank0 -> all): every weight of
 * a handle (decoder, or encoder for an s2v_vae_enc_create handle) lives in ONE device range; the receiving rank copies into
 * it and calls s2v_vae_mark_weights_loaded instead of s2v_vae_load_weight + s2v_vae_finalize.  Stands where the reference
 * would call .to(device) on every rank after from_pretrained (src/inference.py:191-215). */
S2V_API int s2v_vae_weight_arena(s2v_vae* vae, void** dev_ptr, int64_t* bytes);
S2V_API int s2v_vae_mark_weights_loaded(s2v_vae* vae);
/* workspace sets (= tiles in flight) and bytes per set that the last tiled decode ran with: the count follows the free HBM
 * and a byte cap of a quarter of the device's memory (S2V_VAE_WORKSPACE_MAX_GB overrides, <= 0 lifts it); at most six
 * (S2V_VAE_TILES_IN_FLIGHT overrides); for reporting */
S2V_API int s2v_vae_workspace_info(s2v_vae* vae, int32_t* sets, int64_t* bytes_per_set);
/* output extent of s2v_vae_decode for latents [1,F,C,h,w] */
S2V_API int s2v_vae_out_shape(s2v_vae* vae, int32_t F, int32_t h, int32_t w, int32_t tiling, int32_t* Fo, int32_t* Ho, int32_t* Wo);
/* CogVideoXPipeline.decode_latents (pipeline_cogvideox.py:346-351) = 1/scaling_factor * latents, then
 * AutoencoderKLCogVideoX.decode (:1259-1282): frame batches (3,2,2,...) with conv_cache, GroupNorm statistics per
 * frame batch (and per tile when tiling != 0: tiled_decode :1374-1455 incl. its raster-order in-place blends).
 * latents [1,F,C,h,w] model dtype (the pipeline's layout); scaled != 0: pipeline latents (multiplied by
 * 1/scaling_factor here), scaled == 0: already z = latents/scaling_factor (the argument of vae.decode);
 * out [1,out_channels,Fo,Ho,Wo] model dtype.
 * Buffers are (re)allocated when a larger geometry is seen for the first time, never otherwise. */
S2V_API int s2v_vae_decode(s2v_vae* vae, const void* latents, int32_t F, int32_t h, int32_t w, int32_t tiling, int32_t scaled,
                   void* out, s2v_stream stream);
/* VideoProcessor.postprocess_video(output_type="np") (video_processor.py:89-113, image_processor.py:227-240):
 * video [C,F,H,W] -> float32 [F,H,W,C], clamp(x/2 + 0.5, 0, 1) */
S2V_API int s2v_vae_postprocess(const void* video, int32_t C, int32_t F, int32_t H, int32_t W, float* out, int32_t dtype,
                        s2v_stream stream);
/* the same followed by export_to_video's frame conversion `(frame * 255).astype(np.uint8)` (utils/export_utils.py:175;
 * src/video_generate.py:65-66): video [C,F,H,W] -> uint8 [F,H,W,C], what the mp4 writer consumes (4x less D2H traffic) */
S2V_API int s2v_vae_postprocess_u8(const void* video, int32_t C, int32_t F, int32_t H, int32_t W, uint8_t* out, int32_t dtype,
                           s2v_stream stream);

/* ---- reference-image encode (the step in front of the denoise loop) ------------------------------------------
 * Replaces pipe.vae.encode(ref_image).latent_dist.sample() of src/video_generate.py:35-37
 * (AutoencoderKLCogVideoX.encode / _encode / tiled_encode, autoencoder_kl_cogvideox.py:1177-1229,1300-1372;
 * CogVideoXEncoder3D :755-814; CogVideoXDownsample3D downsampling.py:322-353; DiagonalGaussianDistribution
 * autoencoders/vae.py:767-790).  ONE frame: video encode is outside the path.
 * s2v_vae_enc_create builds an s2v_vae handle that holds the ENCODER; its weights are loaded with s2v_vae_load_weight
 * under the reference's "encoder.*" state-dict keys, then s2v_vae_finalize; s2v_vae_destroy frees it.
 * cfg: same struct as the decoder (out_channels = image channels 3, latent_channels = 16). */
S2V_API int s2v_vae_enc_create(const s2v_vae_config* cfg, s2v_vae** out);
/* latent extent of s2v_vae_encode for an H x W image */
S2V_API int s2v_vae_encode_shape(s2v_vae* enc, int32_t H, int32_t W, int32_t tiling, int32_t* h, int32_t* w);
/* image [3,1,H,W] (model dtype, values in [-1,1]) -> moments [2*latent_channels,1,h,w] = the `parameters` of the
 * reference's DiagonalGaussianDistribution (mean | logvar); tiling != 0 follows tiled_encode when the image exceeds
 * (sample_height/2, sample_width/2). */
S2V_API int s2v_vae_encode(s2v_vae* enc, const void* image, int32_t H, int32_t W, int32_t tiling, void* moments, s2v_stream stream);
/* DiagonalGaussianDistribution.sample with the caller's randn: out[c,i] = mean + exp(0.5 * clamp(logvar, -30, 20)) * noise,
 * every operation rounded to `dtype` like the reference's tensor ops; moments [2*C, n_spatial], noise / out [C, n_spatial] */
S2V_API int s2v_vae_gaussian_sample(const void* moments, const void* noise, int32_t latent_channels, int64_t n_spatial, void* out,
                            int32_t dtype, s2v_stream stream);

/* ---- prompt embeddings: T5 v1.1 encoder (the other caller-side step in front of the denoise loop) ----------------
 * Replaces `self.text_encoder(text_input_ids.to(device))[0]` of pipelines/cogvideo/pipeline_cogvideox.py:227 for the
 * T5EncoderModel src/inference.py:183-187 loads.  The arithmetic is transformers' models/t5/modeling_t5.py (third party,
 * not in the reference tree; pinned by tests/golden/t5_tiny.npz): T5LayerNorm, T5Attention (no scaling, relative-position
 * bias of block 0 added to every block's scores, softmax in fp32), T5DenseGatedActDense (gelu_new), no attention mask.
 * Weight names are the HF state-dict keys ("shared.weight", "encoder.block.3.layer.0.SelfAttention.q.weight",
 * "encoder.block.3.layer.1.DenseReluDense.wi_0.weight", "encoder.final_layer_norm.weight", ...). */
typedef struct s2v_t5 s2v_t5;
typedef struct s2v_t5_config {
    int32_t vocab_size;                        /* 32128 (+ added special tokens, src/inference.py:180-189) */
    int32_t d_model, d_kv, num_heads, d_ff;    /* 4096, 64, 64, 10240 */
    int32_t num_layers;                        /* 24 */
    int32_t relative_attention_num_buckets;    /* 32 */
    int32_t relative_attention_max_distance;   /* 128 */
    int32_t dtype;                             /* S2V_DTYPE_* */
    int32_t force_simple;                      /* 1 = generic kernels only (cross-check) */
    float layer_norm_epsilon;                  /* 1e-6 */
    int32_t reserved[5];
} s2v_t5_config;
S2V_API int s2v_t5_create(const s2v_t5_config* cfg, s2v_t5** out);
S2V_API void s2v_t5_destroy(s2v_t5* t5);
S2V_API int s2v_t5_load_weight(s2v_t5* t5, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim,
                       int32_t src_dtype, s2v_stream stream);
S2V_API int s2v_t5_finalize(s2v_t5* t5);
/* the same replica hand-off for the text encoder (9.4 GB in bf16) */
S2V_API int s2v_t5_weight_arena(s2v_t5* t5, void** dev_ptr, int64_t* bytes);
S2V_API int s2v_t5_mark_weights_loaded(s2v_t5* t5);
/* device address of the loaded block-0 relative_attention_bias table [num_buckets, num_heads] (model dtype) */
S2V_API int s2v_t5_rel_table(s2v_t5* t5, void** dev_ptr);
/* position_bias [num_heads, T, T] (model dtype) = T5Attention.compute_bias(T, T) of block 0, gathered by the host from
 * s2v_t5_rel_table with the implementation's own bucket function; sizes the workspace for (B, T) */
S2V_API int s2v_t5_set_position_bias(s2v_t5* t5, const void* bias_dev, int32_t B, int32_t T, s2v_stream stream);
/* input_ids int64 [B, T] -> last_hidden_state [B, T, d_model] (model dtype) */
S2V_API int s2v_t5_encode(s2v_t5* t5, const int64_t* input_ids_dev, int32_t B, int32_t T, void* out, s2v_stream stream);

/* ---- operator-level entry points (used by the parity tests and micro-benchmarks) ------------------------- */
/* C[M,N] = A[M,K] . W[N,K]^T + bias, epilogue 0 = bias, 1 = bias + GELU(tanh); impl 0 = MFMA bf16, 1 = generic (VALU),
 * 3 = fp32 operands on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32; what the fp32 engine runs, bit-identical to impl 1; the launcher picks the
 *     tile; 30 .. 33 force 128 x 128 / 128 x 64 / 64 x 128 / 64 x 64, all with the same bits),
 * 4 = fp16 operands on v_mfma_f32_32x32x16_f16 (what the fp16 engine runs; M, N multiples of 128),
 * 2 = MFMA bf16 with K split over several workgroups per output tile, as the engine runs GEMMs with few tiles and a long
 * reduction (M, N multiples of 256; fails if the shape does not qualify; allocates its workspace, synchronous) */
S2V_API int s2v_op_linear(const void* A, const void* W, const void* bias, void* C, int32_t M, int32_t N, int32_t K,
                  int32_t epilogue, int32_t dtype, int32_t impl, s2v_stream stream);
/* qkv [B*Ntok (+64 rows of slack), 3*H*64] -> out [B*Ntok, H*64]; impl 0 = MFMA flash kernel (needs vt scratch
 * [B*H*64*ceil64(Ntok)] bf16, zero-filled by the caller), 1 = generic */
/* W8A8 linear on the fp8 matrix cores (BASELINE configs[4]: "fp8 (CDNA4 fp8 MFMA) weights"): A [M,K] and W [N,K] bf16 are
 * quantised per row to OCP e4m3 with fp32 scales amax/448 (per token / per output channel) into `scratch`
 * (>= M*K + N*K + 4*(M+N) bytes), multiplied with v_mfma_scale_f32_32x32x64_f8f6f4 and de-quantised in the epilogue:
 * C = epilogue(scale_a[m] * scale_w[n] * acc + bias), bf16.  The reference has no fp8 path (diffusers quantizers are
 * bitsandbytes-only): the contract is stated against this library's own bf16 path (tests/test_gpu_fp8.py).
 * M, N multiples of 256, K a multiple of 128; epilogue 0 = bias, 1 = bias + GELU(tanh). */
S2V_API int s2v_op_linear_fp8(const void* A, const void* W, const void* bias, void* C, int32_t M, int32_t N, int32_t K,
                              int32_t epilogue, void* scratch, int64_t scratch_bytes, s2v_stream stream);
/* FeedForward (attention.py:1237-1243: Linear D -> F, GELU(tanh), Linear F -> D) on the fp8 matrix cores as the fp8 engine runs it:
 * x [M, D], W1 [F, D], W2 [D, F] bf16, quantised per row to e4m3; mx = 1: GELU(h) leaves the first GEMM's epilogue as MX e4m3 (one
 * power-of-two scale per 32 columns, the block scales v_mfma_scale_f32_32x32x64_f8f6f4 takes per lane) and the second GEMM reads it
 * in place; mx = 0: bf16 h + a per-row quantisation pass.  M, D, F multiples of 256; allocates its scratch, synchronous.  No
 * reference arithmetic exists for fp8 (parity unpinned): tests/test_gpu_fp8.py states the contract against a torch emulation. */
S2V_API int s2v_op_ff_fp8(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* out, int32_t M,
                          int32_t D, int32_t F, int32_t mx, s2v_stream stream);
/* out[b][r] = silu(emb[b]) . W[r] + bias[r] for the stacked AdaLN modulation linears of a step (every
 * CogVideoXLayerNormZero.linear and norm_out.linear on silu(temb), normalization.py / cogvideox_transformer_3d.py:122-186);
 * emb [B, temb_dim], W [rows, temb_dim], out [B, rows], B <= 4; impl 0 = the product dispatch, 1 = one wave per row */
S2V_API int s2v_op_mod_gemv(const void* emb, const void* W, const void* bias, void* out, int32_t B, int32_t temb_dim,
                            int64_t rows, int32_t dtype, int32_t impl, s2v_stream stream);
/* Census of the deferred-maximum slow path of the four-wave attention kernels since the last reset (device counters; the call synchronises):
 * slow = slow paths taken, total = (wave, KV tile) pairs run.  attn_p_format 1 lowers the threshold from 2^64 to 2^14; a caller whose data
 * take the slow path in more than a fraction of a percent of the pairs switches back with s2v_set_attn_p_format (the engine's "auto"
 * policy, engine.py, does that after the first step).  s2v_set_attn_p_format drops a captured step; it is re-captured at its next use. */
S2V_API int s2v_attn_slow_stats(s2v_ctx* ctx, uint64_t* slow, uint64_t* total, int32_t reset);
S2V_API int s2v_set_attn_p_format(s2v_ctx* ctx, int32_t attn_p_format);
S2V_API int s2v_op_attention(const void* qkv, void* vt_scratch, void* out, int32_t B, int32_t H, int32_t Ntok, int32_t dtype,
                     int32_t impl, s2v_stream stream);   /* impl: 0 product dispatch (bf16), 1 generic (VALU), 3 = 0 with attn_p_format 1, 4 = attn_q4h (fp16 P) at any length, 5 = fp32 / fp16 storage on the fp32 matrix pipe (what the fp32 engine runs), 6 = fp16 on v_mfma_f32_32x32x16_f16 (what the fp16 engine runs; vt scratch as impl 0) */
/* The same joint attention (F.scaled_dot_product_attention at attention_processor.py:2083-2087, head_dim 64, scale 1/8) as weight_format 2
 * runs it: q (times scale * log2 e) and k of the bf16 qkv rows [B*Ntok, 3*H*64] are quantised to MX e4m3 (32-element blocks along the
 * head dimension, E8M0 scales) into `scratch` and QK^T runs on v_mfma_scale_f32_32x32x64_f8f6f4; V^T (vt_scratch: B*H*64*rup(Ntok,64)
 * bf16), softmax and P.V as s2v_op_attention impl 0.  scratch >= B*H*(66*Ntok + 68*rup(Ntok,64)) + 1024 bytes; it holds, 256-byte
 * aligned and in this order, q8 [B][H][Ntok][64], q8s [B][H][Ntok][2], k8 [B][H][Npad][64], k8s [B][H][Npad/64][64] dwords (byte kb of
 * dword hi*32+r = scale of (key 32*kb+r, block hi)).  No reference arithmetic exists for fp8 (parity unpinned). */
S2V_API int s2v_op_attention_fp8qk(const void* qkv, void* vt_scratch, void* scratch, int64_t scratch_bytes, void* out, int32_t B,
                                   int32_t H, int32_t Ntok, s2v_stream stream);

#ifdef __cplusplus
}
#endif
#endif
