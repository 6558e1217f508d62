/*
 * lidarnerf_hip.h — C ABI of liblidarnerf_hip.so, the MI355X (gfx950) implementation of the LiDAR-NeRF
 * train/render hot path: hash-grid / SH / frequency encodings, fused tiny MLP, per-ray compositing, occupancy
 * indexing — forward and backward.
 *
 * Conventions (every entry point):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller unless its name ends in
 *     `_host` (the caller allocates every output, exactly as the reference's pybind layer does, e.g.
 *     lidarnerf/gridencoder/grid.py:60-67, lidarnerf/raymarching/raymarching.py:235-245);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels are enqueued, never synchronised;
 *   - returns LNH_OK (0) or a negative LNH_ERR_*; never throws, never allocates device memory;
 *     lnh_last_error() returns a thread-local message for the last failure (the reference raises TORCH_CHECK /
 *     std::runtime_error for the same conditions, e.g. gridencoder.cu:430,476,608-624);
 *   - re-entrant: no global state (the reference's ffmlp keeps static stream/event vectors, ffmlp.cu:1020-1049).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference repo root).
 */
#ifndef LIDARNERF_HIP_H
#define LIDARNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNH_API __attribute__((visibility("default")))

typedef void *lnh_stream_t;

enum { LNH_OK = 0, LNH_ERR_INVALID_ARG = -1, LNH_ERR_UNSUPPORTED = -2, LNH_ERR_LAUNCH = -3 };

/* element type of tables / activations */
enum { LNH_F32 = 0, LNH_F16 = 1 };

/* activations, numbering of lidarnerf/ffmlp/ffmlp.py:170-184 (convert_activation) */
enum {
    LNH_ACT_RELU = 0, LNH_ACT_EXPONENTIAL = 1, LNH_ACT_SINE = 2, LNH_ACT_SIGMOID = 3,
    LNH_ACT_SQUAREPLUS = 4, LNH_ACT_SOFTPLUS = 5, LNH_ACT_NONE = 6
};

#define LNH_MAX_LEVELS 32

LNH_API int lnh_version(void);
LNH_API const char *lnh_last_error(void);
/* "gfx950" — the only architecture this library carries code for */
LNH_API const char *lnh_arch(void);
/* "product" for the shipped library; timing-probe builds (tools/probe_variants.py) report their own name here and the
 * Python side refuses to load them unless LNH_ALLOW_VARIANT=1 — results of such builds are wrong by construction. */
LNH_API const char *lnh_build_variant(void);

/* ------------------------------------------------------------------ hash / tiled grid encoder --------------- */
/*
 * Replaces grid_encode_forward   lidarnerf/gridencoder/src/gridencoder.h:12-25 (gridencoder.cu:594-637).
 * inputs [B,D] f32 in [0,1]; embeddings [rows,C] (dtype); offsets_host [L+1] int32 ON THE HOST (the reference
 * passes a device tensor that never changes after construction: grid.py:179-193; a binding caches offsets.cpu());
 * outputs [L,B,C] (dtype) — level-major exactly like the reference (gridencoder.cu:437-438);
 * dy_dx NULL or [B,L,D,C] (dtype).  D in {2,3,4,5}, C in {1,2,4,8}; gridtype 0=hash 1=tiled; interp 0=linear
 * 1=smoothstep.  S = log2(per_level_scale), H = base resolution.
 */
LNH_API int lnh_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host,
                                    void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                    uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp,
                                    int dtype, lnh_stream_t stream);
/*
 * Replaces grid_encode_backward  gridencoder.h:26-41 (gridencoder.cu:639-693).
 * grad [L,B,C] (dtype); grad_embeddings [rows,C] (dtype), ZERO-INITIALISED by the caller (grid.py:106), receives
 * atomic scatter-adds; dy_dx / grad_inputs NULL or [B,L,D,C] / [B,D] (dtype).  `embeddings` is unused by the
 * arithmetic (kept for signature parity, may be NULL).
 */
LNH_API int lnh_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                                     const int32_t *offsets_host, void *grad_embeddings, uint32_t B, uint32_t D,
                                     uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs,
                                     uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                     lnh_stream_t stream);
/*
 * Same result as lnh_grid_encode_backward for the hot configuration (D == 3, C == 2) without a single atomic add to
 * HBM: contributions are binned per 8192-row table bucket in a caller-provided workspace and reduced in LDS in 64-bit
 * fixed point (grid.hip, "bucketed backward"): the result is one integer sum per table row, independent of execution
 * order (bit-reproducible run to run), rounded once to the table type.  `workspace` is scratch device memory of at least
 * lnh_grid_backward_workspace_size(...) bytes (0 = configuration not supported by this path); its content is
 * irrelevant before and after the call.  No dy_dx / grad_inputs (LiDAR sample positions carry no gradient).
 */
LNH_API uint64_t lnh_grid_backward_workspace_size(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C,
                                                  uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                                  int align_corners, int dtype);
/* The workspace serves one chunk of the batch at a time: any size between lnh_grid_backward_workspace_size_min(...) and
 * lnh_grid_backward_workspace_size(...) is accepted — a smaller workspace means shorter chunks (more launches: 3.09 / 1.60 /
 * 0.91 GB cost 897 / 922 / 992 us at 4096 rays x 832 samples), never another result class.  Below the minimum the call
 * returns LNH_ERR_INVALID_ARG and lnh_last_error() names the minimum. */
LNH_API uint64_t lnh_grid_backward_workspace_size_min(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C,
                                                      uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                                      int align_corners, int dtype);
/* Host-side description of that workspace's bucket plan for `level` (no device work): out[0] = table buckets of the
 * level, out[1] = pool slots per bucket (what exceeds them goes to the level's spill list), out[2] = rows per bucket,
 * out[3] = entries per reduce slice (a bucket with more is reduced by several workgroups).  The sum the call produces
 * never depends on these numbers — integer accumulation (see grid.hip) — the tests use them to build inputs that
 * overflow a bucket by a few entries or split one.  Returns LNH_ERR_UNSUPPORTED where workspace_size returns 0. */
/* Testing knob, process-wide: entries per reduce slice (0 restores the default, 512 K; clamped to [1024, 512 K]) — with the
 * default only a concentrated batch of more than half a million entries per bucket is reduced in slices; a small value
 * lets the tests reach that path with small inputs.  Set it BEFORE lnh_grid_backward_workspace_size (more slices need
 * more image slots).  Never changes a result. */
LNH_API void lnh_grid_backward_set_slice_entries(uint32_t entries);
LNH_API int lnh_grid_backward_plan_info(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                        float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                        uint32_t level, uint32_t *out4);
LNH_API int lnh_grid_encode_backward_ws(const void *grad, const float *inputs, const int32_t *offsets_host,
                                        void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                        uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                        void *workspace, uint64_t workspace_bytes, lnh_stream_t stream);
/* Same, restricted to the levels [level_begin, level_end): rows of other levels are not touched.  Lets a data-parallel
 * caller hand the gradient of finished levels to the all-reduce while later levels are still being reduced. */
LNH_API int lnh_grid_encode_backward_ws_levels(const void *grad, const float *inputs, const int32_t *offsets_host,
                                               void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                               float S, uint32_t H, uint32_t gridtype, int align_corners,
                                               uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes,
                                               uint32_t level_begin, uint32_t level_end, lnh_stream_t stream);
/* The same backward in two steps for a data-parallel caller (bit-identical result): `_begin` runs everything except the
 * reduce pass of the (last) chunk; `_finish` runs that reduce pass for the levels [level_begin, level_end), after which
 * their rows of grad_embeddings are final.  Call `_begin` once, then `_finish` for consecutive level windows covering
 * [0, L), with the same arguments and the same workspace on the same stream: the gradient of a finished window can go to
 * the all-reduce while the next window is being reduced, and the scatter pass is NOT cut into windows (which costs it a
 * quarter of its speed). */
LNH_API int lnh_grid_encode_backward_ws_begin(const void *grad, const float *inputs, const int32_t *offsets_host,
                                              void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                              float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                              int dtype, void *workspace, uint64_t workspace_bytes, lnh_stream_t stream);
LNH_API int lnh_grid_encode_backward_ws_finish(const void *grad, const float *inputs, const int32_t *offsets_host,
                                               void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                               float S, uint32_t H, uint32_t gridtype, int align_corners,
                                               uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes,
                                               uint32_t level_begin, uint32_t level_end, lnh_stream_t stream);
/* The same entry points behind ONE signature, for a caller that has just cleared its buffers itself (a training step that
 * clears every accumulated-into buffer of its backward pass with one lnh_zero_regions launch):
 *   split  0 = lnh_grid_encode_backward_ws_levels, 1 = ..._begin (level_begin / level_end ignored), 2 = ..._finish
 *   flags  LNH_BWD_WS_CLEARED: the first lnh_grid_backward_workspace_clear_bytes(...) bytes of `workspace` are zero on
 *          entry (the cursors of the batch's FIRST chunk: that chunk's clear launch is skipped);
 *          LNH_BWD_TABLE_ZERO: grad_embeddings holds zeros on entry — the reduce pass of the first chunk stores its sums
 *          instead of adding them to rows it would have to read first (the same values: 0 + x).
 * A caller that sets a flag without having cleared gets garbage; without flags this is exactly the entry point `split` names.
 * lnh_grid_backward_workspace_clear_bytes: size of that head for the chunk plan the call will choose for `workspace_bytes`
 * (0: unsupported configuration / workspace too small). */
#define LNH_BWD_WS_CLEARED 1u
#define LNH_BWD_TABLE_ZERO 2u
LNH_API int lnh_grid_encode_backward_ws_ex(const void *grad, const float *inputs, const int32_t *offsets_host,
                                           void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                           uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                           void *workspace, uint64_t workspace_bytes, uint32_t level_begin,
                                           uint32_t level_end, int split, uint32_t flags, lnh_stream_t stream);
LNH_API uint64_t lnh_grid_backward_workspace_clear_bytes(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C,
                                                         uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                                         int align_corners, uint32_t interp, int dtype,
                                                         uint64_t workspace_bytes);
/*
 * Replaces grad_total_variation  gridencoder.h:43-55 (gridencoder.cu:695-910): adds the TV-regulariser gradient
 * of the cells visited by `inputs` into `grad` (same layout as embeddings).
 */
LNH_API int lnh_grad_total_variation(const void *inputs, const void *embeddings, void *grad,
                                     const int32_t *offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C,
                                     uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                     lnh_stream_t stream);
/*
 * Bit-exact contract of get_grid_index / fast_hash (gridencoder.cu:53-93): element index (row*C) of the 2^D
 * corners of every (level, point); out_idx [L,B,2^D] uint32, 0xffffffff for out-of-range points.
 */
LNH_API int lnh_grid_corner_indices(const float *inputs, const int32_t *offsets_host, uint32_t *out_idx, uint32_t B,
                                    uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                    int align_corners, lnh_stream_t stream);

/* ------------------------------------------------------------------ frequency encoder ----------------------- */
/* Replaces freq_encode_forward   lidarnerf/freqencoder/src/freqencoder.h:8-14 (freqencoder.cu:103-122).
 * inputs [B,D] f32 -> outputs [B,C] f32, C = D + 2*D*deg, layout [x | sin(2^f x) | cos(2^f x)]_f. */
LNH_API int lnh_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                    float *outputs, lnh_stream_t stream);
/* Replaces freq_encode_backward  freqencoder.h:16-22 (freqencoder.cu:124-147); uses the SAVED outputs. */
LNH_API int lnh_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg,
                                     uint32_t C, float *grad_inputs, lnh_stream_t stream);

/* ------------------------------------------------------------------ spherical-harmonics encoder -------------- */
/* Replaces sh_encode_forward     lidarnerf/shencoder/src/shencoder.h:9-14 (shencoder.cu:887-912).
 * inputs [B,3] f32 (raw direction) -> outputs [B,degree^2] f32; dy_dx NULL or [B,3,degree^2]. degree in 1..4. */
LNH_API int lnh_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree,
                                  float *dy_dx, lnh_stream_t stream);
/* Replaces sh_encode_backward    shencoder.h:15-20 (shencoder.cu:914-943): grad_inputs[b,d] += sum grad*dy_dx. */
LNH_API int lnh_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t degree,
                                   const float *dy_dx, float *grad_inputs, lnh_stream_t stream);

/* ------------------------------------------------------------------ fully fused MLP -------------------------- */
/*
 * Replaces ffmlp_forward / ffmlp_inference / ffmlp_backward  lidarnerf/ffmlp/src/ffmlp.h:7-44
 * (ffmlp.cu:866-1263) and the bias-free Linear stacks of lidarnerf/nerf/network.py:45-99.
 * inputs [B,input_dim] f16; weights flat f16: [hidden*input | hidden*hidden*n_hidden_mats | output_dim*hidden],
 * every matrix row-major [out,in] (ffmlp.py:222-226 with n_hidden_mats = num_layers-1; n_hidden_mats = 0 gives
 * the 2-matrix sigma net).  output_dim is the PADDED width (multiple of 16, <= 16 for the fused path);
 * outputs [B,output_dim] f16.  forward_buffer NULL (inference) or [n_hidden_mats+1, B, hidden] f16 receiving the
 * post-activation of every hidden layer (ffmlp.py:43-58).  Accumulation is fp32 on MFMA (the reference
 * accumulates in fp16 WMMA fragments, ffmlp.cu:788-791).  B must be a multiple of 16.
 */
LNH_API int lnh_mlp_forward(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim,
                            uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats, uint32_t activation,
                            uint32_t output_activation, void *forward_buffer, void *outputs, lnh_stream_t stream);
/*
 * Weight gradients of the MLP family are FIXED-ORDER sums: every backward kernel leaves one partial sum per workgroup in
 * `wgrad_ws` and a second, small launch on the same stream adds the partials up in workgroup-index order and adds the result
 * to the gradient (csrc/wgrad.h) — the same bits on every run, as with the reference's split-K GEMMs (ffmlp.cu:1107-1142);
 * rounds 1-5 used fp32 device atomics, whose order changed from run to run.
 * wgrad_ws: device scratch of lnh_wgrad_workspace_bytes() bytes, 16-byte aligned, contents irrelevant; one workspace serves
 * all launches of a stream, launches that may overlap in time (several streams) need one each.  Taken by lnh_mlp_backward,
 * lnh_density_mlp_backward, lnh_lidar_color_backward(_image), lnh_ragged_color_backward, lnh_lidar_dir_term_backward and
 * their _bf16 builds.
 */
LNH_API uint64_t lnh_wgrad_workspace_bytes(void);
/*
 * grad [B,output_dim] f16; grad_weights: f32 flat vector (same ordering as weights, 16-byte aligned): the batch's sum is
 * ADDED to it (clear it first for a plain gradient); grad_inputs NULL or [B,input_dim] f16.
 * The hidden activations are recomputed from `inputs` (no forward_buffer needed).
 */
LNH_API int lnh_mlp_backward(const void *grad, const void *inputs, const void *weights, uint32_t B,
                             uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats,
                             uint32_t activation, uint32_t output_activation, void *grad_inputs, float *grad_weights,
                             void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);

/*
 * Shapes.  hidden_dim 32 / 64 with n_hidden_mats <= 2: register-resident kernels, and lnh_mlp_backward is ONE kernel.
 * hidden_dim 128 / 256 (ffmlp.cu:756-800) and n_hidden_mats 3 .. 14 at any of the four widths: lnh_mlp_forward runs
 * kernels that load a weight fragment where it is used (csrc/mlp_wide.hip); the backward is the reference's own split
 * (ffmlp.cu:578-733 fused activation gradients + 1107-1263 split-K GEMMs for the weights):
 *   lnh_mlp_backward_data: from grad [B,output_dim], the forward_buffer lnh_mlp_forward filled, and weights_t — the
 *     matrices TRANSPOSED, flat [input*hidden (W0^T, rows = inputs) | hidden*hidden*n_hidden_mats (each Wh^T) |
 *     hidden*output_dim (Wo^T, rows = hidden units)] — writes backward_buffer [n_hidden_mats+1, B, hidden] f16 =
 *     dL/d(pre-activation) of every hidden layer and grad_inputs (NULL or [B,input_dim]);
 *   the weight gradients are the GEMMs dW0 = backward_buffer[0]^T inputs, dWh_m = backward_buffer[m+1]^T
 *     forward_buffer[m], dWo = grad^T forward_buffer[n_hidden_mats]: one lnh_mlp_wgrad call each (below).
 *   lnh_mlp_backward returns LNH_ERR_UNSUPPORTED for these shapes and says so.  Works for the narrow shapes too.
 * hidden_dim 16 has no kernel of its own (the module zero-pads it onto 32); input_dim > 128: LNH_ERR_UNSUPPORTED.
 */
LNH_API int lnh_mlp_backward_data(const void *grad, const void *forward_buffer, const void *weights_t, uint32_t B,
                                  uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats,
                                  uint32_t activation, void *backward_buffer, void *grad_inputs, lnh_stream_t stream);
/*
 * Weight gradient of ONE layer of a wide fused MLP, the contraction over the batch the reference runs as a split-K CUTLASS
 * GEMM (ffmlp.cu:1107-1263, cutlass_matmul.h:481-616):  grad_weights[M, N] += grad^T acts,  grad [B, M] and acts [B, N] f16
 * (row-major; rows of lnh_mlp_backward_data's backward_buffer / lnh_mlp_forward's forward_buffer / the MLP input), M and N
 * multiples of 16 in 16 .. 256, grad_weights f32 row-major (a slice of the flat gradient vector), 16-byte aligned pointers.
 * Fixed-order sum over the batch (wgrad_ws: lnh_wgrad_workspace_bytes): the same bits on every run.
 */
LNH_API int lnh_mlp_wgrad(const void *grad, const void *acts, uint32_t B, uint32_t M, uint32_t N, float *grad_weights,
                          void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);

/* ------------------------------------------------------------------ ray utilities / occupancy grid ----------- */
/* Replaces near_far_from_aabb    lidarnerf/raymarching/src/raymarching.h:6-12 (raymarching.cu:104-177). */
LNH_API int lnh_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                                   float min_near, float *nears, float *fars, lnh_stream_t stream);
/* What NeRFRenderer.run_cuda does in front of the marcher, in ONE launch (no pybind counterpart: the reference's
 * renderer does it with torch ops — renderer.py:129-138 for the LiDAR range; this build's run_cuda cuts it at the box):
 *   nears[n] = near;  fars[n] = torch.minimum(near * far_factor, far of lnh_near_far_from_aabb(rays_o, rays_d, aabb, near))
 * (bit for bit, NaN exits included), and up to 4 device regions cleared (the marcher's zero-initialised sample buffers —
 * raymarching.py:235-245 —, its counter, the colour buffer): host arrays of pointers and byte counts, 4-byte granular. */
LNH_API int lnh_lidar_march_prologue(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float near,
                                     float far_factor, float *nears, float *fars, void *const *zero_ptrs,
                                     const uint64_t *zero_bytes, uint32_t zero_count, lnh_stream_t stream);
/* Replaces sph_from_ray          raymarching.h:13-17 (raymarching.cu:182-231). */
LNH_API int lnh_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                             lnh_stream_t stream);
/* Replaces morton3D / morton3D_invert  raymarching.h:18-21 (raymarching.cu:71-95,237-279) — bit exact. */
LNH_API int lnh_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, lnh_stream_t stream);
LNH_API int lnh_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, lnh_stream_t stream);
/* Replaces packbits              raymarching.h:22-25 (raymarching.cu:286-319): N bytes from 8N floats. */
LNH_API int lnh_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, lnh_stream_t stream);
/* Cell index / occupancy bit of arbitrary points (the lookup inside kernel_march_rays_train,
 * raymarching.cu:386-408) — exposes the bit-exact indexing contract on its own. */
LNH_API int lnh_occupancy_lookup(const float *xyz, const float *dt, const uint8_t *bitfield, float bound, uint32_t N,
                                 uint32_t C, uint32_t H, uint32_t *cell_index, uint8_t *occ, lnh_stream_t stream);
/* Replaces march_rays_train      raymarching.h:27-44 (raymarching.cu:331-568).  counter [2] int32 zeroed by the
 * caller: [0] += points, [1] += rays; rays [N,3] = (ray id, offset, count) in atomic-allocation order. */
LNH_API int lnh_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                 const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                                 int32_t *rays, int32_t *counter, const float *noises, lnh_stream_t stream);
/* Replaces composite_rays_train_forward / _backward  raymarching.h:45-69 (raymarching.cu:577-802). */
LNH_API int lnh_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                             const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                             float *weights_sum, float *depth, float *image, lnh_stream_t stream);
LNH_API int lnh_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                              const float *sigmas, const float *rgbs, const float *deltas,
                                              const int32_t *rays, const float *weights_sum, const float *image,
                                              uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                              float *grad_rgbs, lnh_stream_t stream);
/*
 * LiDAR variant of the ragged compositing (no reference counterpart: the reference composites LiDAR rays with PyTorch
 * ops on dense [N,T] tensors, renderer.py:233-271, and its CUDA template above has 3 colour channels and no depth
 * gradient, raymarching.py:330).  feats [M,K] (K <= 3; K = 2: ray-drop, intensity), deltas [M,2] and xyzs [M,3] as
 * written by lnh_march_rays_train; depth = sum w * z with z = (xyz - o) . d the ABSOLUTE distance along the unit ray.
 * Backward: grad_sigmas [M] / grad_feats [M,K] ZERO-INITIALISED by the caller, includes the depth term.
 */
LNH_API int lnh_lidar_composite_rays_train_forward(const float *sigmas, const float *feats, const float *deltas,
                                                   const float *xyzs, const float *rays_o, const float *rays_d,
                                                   const int32_t *rays, uint32_t M, uint32_t N, uint32_t K,
                                                   float T_thresh, float *weights_sum, float *depth, float *image,
                                                   lnh_stream_t stream);
LNH_API int lnh_lidar_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_depth,
                                                    const float *grad_image, const float *sigmas, const float *feats,
                                                    const float *deltas, const float *xyzs, const float *rays_o,
                                                    const float *rays_d, const int32_t *rays,
                                                    const float *weights_sum, const float *depth, const float *image,
                                                    uint32_t M, uint32_t N, uint32_t K, float T_thresh,
                                                    float *grad_sigmas, float *grad_feats, lnh_stream_t stream);

/*
 * Inference variants (raymarching.h:55-69 march_rays / composite_rays; raymarching.cu:808-928, 966-1053): march the
 * first n_alive rays listed in rays_alive for at most n_step occupied samples from their current rays_t (outputs
 * [n_alive*n_step, 3|3|2], caller-zeroed: delta == 0 ends a ray), then accumulate sigmas / rgbs [n_alive*n_step, 1|3]
 * into weights_sum / depth / image [N, 1|1|3] in place; finished rays get rays_alive[n] = -1, the others their new t.
 */
LNH_API int lnh_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                           const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                           uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                           float *xyzs, float *dirs, float *deltas, const float *noises, lnh_stream_t stream);
LNH_API int lnh_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                               const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum,
                               float *depth, float *image, lnh_stream_t stream);

/* ------------------------------------------------------------------ LiDAR renderer kernels ------------------ */
/*
 * The reference composites LiDAR rays with ~40 PyTorch launches (lidarnerf/nerf/renderer.py:180-271).  These
 * entry points are the same arithmetic as single kernels, one 64-lane wavefront per ray.
 *
 * lnh_lidar_weights: renderer.py:233-243 (and 180-194).  z [N,T] sorted per ray, sigma [N,T], sample_dist [N];
 *   deltas_i = z_{i+1}-z_i (last = sample_dist), alpha = 1-exp(-delta*density_scale*sigma),
 *   w = alpha * prod_{j<i}(1-alpha_j+1e-15).  Writes weights [N,T] f32.
 */
LNH_API int lnh_lidar_weights(const float *z, const float *sigma, const float *sample_dist, uint32_t N, uint32_t T,
                              float density_scale, float *weights, lnh_stream_t stream);
/*
 * lnh_lidar_composite_forward: renderer.py:233-271.  rgb [N,T,K] f32 (already zero where weights <= 1e-4,
 * network.py:204-208).  Outputs weights [N,T], weights_sum [N], depth [N] = sum w*z, image [N,K] = sum w*rgb.
 */
LNH_API int lnh_lidar_composite_forward(const float *z, const float *sigma, const float *rgb,
                                        const float *sample_dist, uint32_t N, uint32_t T, uint32_t K,
                                        float density_scale, float *weights, float *weights_sum, float *depth,
                                        float *image, lnh_stream_t stream);
/*
 * lnh_lidar_composite_backward: exact adjoint of the PyTorch graph above (autograd of cumprod / exp / sums),
 * INCLUDING the depth gradient (the reference's CUDA composite drops it, raymarching.py:330; the LiDAR loss is
 * depth dominated so the PyTorch path is the one to follow).  grad_* of the three outputs -> grad_sigma [N,T],
 * grad_rgb [N,T,K].
 */
LNH_API int lnh_lidar_composite_backward(const float *grad_weights_sum, const float *grad_depth,
                                         const float *grad_image, const float *z, const float *sigma,
                                         const float *rgb, const float *sample_dist, uint32_t N, uint32_t T,
                                         uint32_t K, float density_scale, float *grad_sigma, float *grad_rgb,
                                         lnh_stream_t stream);
/*
 * lnh_lidar_resample: renderer.py:180-231 in one kernel — stage-1 weights, sample_pdf (renderer.py:10-46) with
 * the caller's uniforms u [N,n_new] (linspace for det, rand for training), then the sort/merge of the old and new
 * z values.  Outputs new_z [N,n_new] (sorted_new = 0: in sample_pdf's order, exactly what the reference hands to
 * its second density query; sorted_new = 1: ascending, same set of values), merged z_out [N,T+n_new] ascending
 * and perm [N,T+n_new] int32: the position in concat([old, new]) each merged element came from (= torch.sort's
 * index, renderer.py:218).
 */
LNH_API int lnh_lidar_resample(const float *z, const float *sigma, const float *sample_dist, const float *u,
                               uint32_t N, uint32_t T, uint32_t n_new, float density_scale, uint32_t sorted_new,
                               float *new_z, float *z_out, int32_t *perm, lnh_stream_t stream);
/* The same with the rows of `sigma` sigma_stride (>= T) floats apart: the fused render step keeps the coarse and the
 * importance samples of a ray side by side in one [N, T+n_new] buffer and resamples from its first T columns. */
LNH_API int lnh_lidar_resample_strided(const float *z, const float *sigma, uint32_t sigma_stride,
                                       const float *sample_dist, const float *u, uint32_t N, uint32_t T,
                                       uint32_t n_new, float density_scale, uint32_t sorted_new, float *new_z,
                                       float *z_out, int32_t *perm, lnh_stream_t stream);
/* lnh_lidar_resample_strided with sorted_new = 1 that also writes the grid coordinates of the new samples
 * (lnh_lidar_sample_points for slots T .. T+n_new-1) into x01 [N*(T+n_new), 3]: the importance pass of the fused step
 * needs no separate coordinate kernel. */
LNH_API int lnh_lidar_resample_points(const float *z, const float *sigma, uint32_t sigma_stride,
                                      const float *sample_dist, const float *u, uint32_t N, uint32_t T, uint32_t n_new,
                                      float density_scale, float *new_z, float *z_out, int32_t *perm,
                                      const float *rays_o, const float *rays_d, const float *aabb, float bound,
                                      float *x01, lnh_stream_t stream);


/* ------------------------------------------------------------------ fused LiDAR field step ------------------ */
/*
 * The kernels below fuse what lidarnerf/nerf/renderer.py:149-256 + lidarnerf/nerf/network.py:162-237 spread over
 * dozens of PyTorch launches (sample positions, encoder permutes, trunc_exp, sort/gather merge, masked colour
 * query).  Semantics are unchanged; see lidar-nerf_amd/csrc/lidar_field.hip for the derivations.
 *
 * lnh_lidar_sample_points: (clip(o + d*z, aabb) + bound) / (2 bound)  (renderer.py:164-167, grid.py:213) for z [N,T],
 * written to row n*T_tot + slot_off + j of x01 (T_tot = T, slot_off = 0: plain [N*T,3]).
 */
LNH_API int lnh_lidar_sample_points(const float *rays_o, const float *rays_d, const float *z, const float *aabb,
                                    float bound, uint32_t N, uint32_t T, uint32_t T_tot, uint32_t slot_off, float *x01,
                                    lnh_stream_t stream);
/* lnh_lidar_coarse_samples (below: the stratified depths of renderer.py:140-156, u = NULL for the unperturbed ones) and
 * lnh_lidar_sample_points of those depths (slot_off = 0) in one launch; z [N,T] and x01 rows n*T_tot + j are written. */
LNH_API int lnh_lidar_coarse_sample_points(const float *u, const float *rays_o, const float *rays_d, const float *aabb,
                                           float bound, uint32_t N, uint32_t T, uint32_t T_tot, float near, float far,
                                           float *z, float *x01, lnh_stream_t stream);
/*
 * lnh_grid_encode_forward_mapped: lnh_grid_encode_forward (D = 3, hash, linear) whose launch index b = r*T_cur + j
 * reads position row r*T_tot + slot_off + j of inputs_all [B_all,3] and writes the same row of
 * outputs_all [L, B_all, C] — coarse and importance samples of a ray share one buffer, so the backward pass is ONE
 * launch over B_all points.
 */
LNH_API int lnh_grid_encode_forward_mapped(const float *inputs_all, const void *embeddings, const int32_t *offsets_host,
                                           void *outputs_all, uint32_t B, uint32_t T_cur, uint32_t T_tot,
                                           uint32_t slot_off, uint32_t B_all, uint32_t C, uint32_t L, float S,
                                           uint32_t H, int dtype, lnh_stream_t stream);
/*
 * lnh_density_mlp_forward: sigma-net 32 -> 64 -> 16 (ReLU, no bias; network.py:45-59,162-179) on features in the
 * encoder's level-major layout [16,B,2] (fp16).  Point p = r*T_cur + j writes row r*T_tot + slot_off + j of
 * h16 [*,16] fp16 (raw outputs: col 0 density pre-activation, cols 1..15 geo_feat) and sigma [*] f32 = exp(h16[.,0])
 * (trunc_exp forward).  weights flat fp16 [64*32 | 16*64].  feat_rows = 0: features is [16,B,2] indexed by p;
 * feat_rows = B_all: features is [16,B_all,2] indexed by the destination row (lnh_grid_encode_forward_mapped).
 */
LNH_API int lnh_density_mlp_forward(const void *features, const void *weights, uint32_t B, uint32_t T_cur,
                                    uint32_t T_tot, uint32_t slot_off, uint32_t feat_rows, void *h16, float *sigma,
                                    lnh_stream_t stream);
/* grad_h16 rows addressed like h16 above -> grad_features [16,B,2] fp16, grad_weights fp32 += (fixed-order sum, see
 * lnh_wgrad_workspace_bytes). */
LNH_API int lnh_density_mlp_backward(const void *grad_h16, const void *features, const void *weights, uint32_t B,
                                     uint32_t T_cur, uint32_t T_tot, uint32_t slot_off, void *grad_features,
                                     float *grad_weights, void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
/*
 * lnh_lidar_merge_weights: sigma_m[n,i] = sigma_pt[n, perm[n,i]] and the compositing weights of the merged samples
 * (renderer.py:217-243); z [N,T] merged (sorted) depths, perm from lnh_lidar_resample.
 */
LNH_API int lnh_lidar_merge_weights(const float *z, const float *sigma_pt, const int32_t *perm,
                                    const float *sample_dist, uint32_t N, uint32_t T, float density_scale,
                                    float *sigma_m, float *weights, lnh_stream_t stream);
/*
 * Element-wise stages around the field that the reference leaves to dozens of PyTorch launches (one launch each here).
 * lnh_lidar_coarse_samples: z[n,i] = near + (far-near)*linspace(0,1,T)[i] (+ (u[n,i]-0.5)*(far-near)/T when u != NULL)
 *   (renderer.py:147-161; linspace evaluated symmetrically like torch.linspace).
 * lnh_lidar_dir_term: per-ray direction term of the colour head's first Linear (network.py:215-221):
 *   features16[n,k] = fp16-rounded dir_features[n,k] (kept as f32), cdir[n,o] = sum_k features16[n,k]*fp16(w0[o*ldw+k]),
 *   o < 64, K <= 128.
 * lnh_lidar_pack_weights: fp32 master matrices (row strides ld_*) -> flat fp16 vectors of lnh_density_mlp_* (wsig16:
 *   [64,32 | 16,64]) and lnh_lidar_color_* (wcol16: W0g [64,16] = (0 | wc0[:, n_dir:n_dir+15]) | wc1 [64,64] | wc2 -> [16,64]).
 * lnh_lidar_loss: nerf/utils.py:712-746 default criteria, loss = mean_n(alpha_d*|d-gd| + alpha_r*(r-gr)^2 +
 *   alpha_i*(i-gi)^2) with predictions/targets masked by gt ray-drop; gt [N,3] = (raydrop, intensity, depth); also
 *   writes d loss/d depth [N] and d loss/d image [N,2], multiplied by *grad_scale when grad_scale != NULL (the loss
 *   scale of the training step: backward() then has nothing left to multiply).  `loss` need not be cleared.
 */
LNH_API int lnh_lidar_coarse_samples(const float *u, uint32_t N, uint32_t T, float near, float far, float *z,
                                     lnh_stream_t stream);
LNH_API int lnh_lidar_dir_term(const float *dir_features, const float *w0, uint32_t ldw, uint32_t N, uint32_t K,
                               float *features16, float *cdir, lnh_stream_t stream);
/* The same with the frequency encoder of the directions folded in: dirs [N,3] -> features [N, 3 + 6*degree]
 * (lnh_freq_encode_forward's layout and arithmetic: x | sin(2^f x), sin(2^f x + pi/2) per band) -> features16, cdir. */
LNH_API int lnh_lidar_dir_term_freq(const float *dirs, uint32_t degree, const float *w0, uint32_t ldw, uint32_t N,
                                    float *features16, float *cdir, lnh_stream_t stream);
/* grad_w0[o*ldw + k] += sum_n ray_sum[n,o] * features16[n,k], k < K  (ray_sum [N,64] from lnh_lidar_color_backward; the
 * sum over the rays is formed in a fixed order — wgrad_ws, see lnh_wgrad_workspace_bytes — and ADDED: clear grad_w0 first).  grad_w0g != NULL: the packed [64,16] gradient of the colour
 * head's geo-feature columns (what lnh_lidar_color_backward accumulates at the front of its grad_w) is copied to
 * grad_w0[o*ldw + K + c] = grad_w0g[o*16 + 1 + c], c < 15 — the whole gradient of network.py:199's first Linear in one launch. */
LNH_API int lnh_lidar_dir_term_backward(const float *ray_sum, const float *features16, uint32_t N, uint32_t K,
                                        const float *grad_w0g, float *grad_w0, uint32_t ldw, void *wgrad_ws,
                                        uint64_t wgrad_ws_bytes, lnh_stream_t stream);
LNH_API int lnh_lidar_pack_weights(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1,
                                   const float *wc0, uint32_t ld_c0, uint32_t n_dir, const float *wc1, uint32_t ld_c1,
                                   const float *wc2, uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream);
/* Everything a fused render step needs before its first encode, in ONE launch: lnh_lidar_pack_weights (n_dir = 3 + 6 *
 * degree) + lnh_lidar_dir_term_freq (on rays_d, against wc0) + lnh_lidar_coarse_sample_points — the same arithmetic, bit
 * for bit (renderer.py:129-167 sampling set-up, network.py:215-221 direction term). */
LNH_API int lnh_lidar_step_prologue(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                                    uint32_t ld_c0, uint32_t degree, const float *wc1, uint32_t ld_c1, const float *wc2,
                                    uint32_t ld_c2, void *wsig16, void *wcol16, const float *u, const float *rays_o,
                                    const float *rays_d, const float *aabb, float bound, uint32_t N, uint32_t T,
                                    uint32_t T_tot, float near, float far, float *z, float *x01, float *features16,
                                    float *cdir, lnh_stream_t stream);
LNH_API int lnh_lidar_loss(const float *depth, const float *image, const float *gt, uint32_t N, float alpha_d,
                           float alpha_r, float alpha_i, const float *grad_scale, float *loss, float *grad_depth,
                           float *grad_image, lnh_stream_t stream);
/* The same with the structural-gradient term of the reference's patch epochs (nerf/utils.py:760-876, non-sobel grad_loss):
 * rays come as N / (px * py) patches of px x py pixels, row-major; + alpha_grad * mean_{patch, row, j < py-1} |
 * |pd_j - pd_j+1| m_j - (gd_j - gd_j+1) m_j |, depths in metres (value / scale), m_j = raydrop_j * (|gd_j - gd_j+1| < 0.01). */
LNH_API int lnh_lidar_loss_patch(const float *depth, const float *image, const float *gt, uint32_t N, uint32_t px,
                                 uint32_t py, float scale, float alpha_d, float alpha_r, float alpha_i, float alpha_grad,
                                 const float *grad_scale, float *loss, float *grad_depth, float *grad_image,
                                 lnh_stream_t stream);
/*
 * The LiDAR colour head on the marcher's RAGGED samples (BASELINE config 4; network.py:199-237 evaluated on the samples of
 * renderer.run_cuda, which the reference dropped while keeping raymarching.cu:331-772), with the dense chain's two moves:
 * the direction part of the first Linear once per RAY (cdir [N,64] from lnh_lidar_dir_term_freq on rays_d) and the
 * sample's sigma-net row as the 16-wide input.  rays [N,3] i32 = the marcher's table (ray index, first sample, count);
 * a ray whose samples do not fit M has none.  w16 = lnh_lidar_pack_weights' wcol16.
 * lnh_ragged_color_forward: rgb[m] = sigmoid(head(h16[m], cdir[ray])) [M,2] for every owned sample (others untouched).
 * lnh_ragged_color_backward: grad_h16 rows of the owned samples (col 0 = grad_sigma * density_scale * exp(clamp(h0)), cols
 *   1..15 through the head; rows nobody owns are NOT written: clear grad_h16 first), grad_w += (layout of wcol16, fp32),
 *   ray_sum [N,64] = sum over the ray's samples of d(first hidden pre-activation), indexed by ray INDEX (feeds
 *   lnh_lidar_dir_term_backward; rows of rays without an entry are not written).
 */
LNH_API int lnh_ragged_color_forward(const void *h16, const int32_t *rays, const float *cdir, const void *w16, uint32_t N,
                                     uint32_t M, float *rgb, lnh_stream_t stream);
LNH_API int lnh_ragged_color_backward(const float *grad_rgb, const float *grad_sigma, float density_scale, const void *h16,
                                      const int32_t *rays, const float *cdir, const void *w16, uint32_t N, uint32_t M,
                                      void *grad_h16, float *grad_w, float *ray_sum, void *wgrad_ws,
                                      uint64_t wgrad_ws_bytes, lnh_stream_t stream);
/*
 * Element-wise stages of the occupancy-grid render chain over the marcher's flat sample list [M] (BASELINE config 4; the
 * reference kept torch-ngp's kernels, raymarching.cu:331-772, and dropped this caller — what runs between them are the
 * tensor expressions of network.py:162-237 on [M, *] tensors: one launch each here).
 * lnh_ragged_points: x01 = (xyz + bound) / (2 bound) (gridencoder/grid.py:213).
 * lnh_ragged_pack_weights: fp32 master matrices -> wsig16 [64,32 | 16,64] and wcol16 [64,96 (wc0[:, :n_in], zero-padded) |
 *   64,64 | 16,64 (wc2 in rows 0..1)] — the flat vectors of lnh_density_mlp_* / lnh_mlp_*(input_dim 96, one hidden matrix).
 * lnh_ragged_color_input: cin [M,96] = [x | sin(2^f x), sin(2^f x + pi/2), f < degree (freqencoder.cu:34-63) of the
 *   sample's direction | geo_feat = h16[m, 1:16] | 0] (network.py:215-221), in the MLP element type.
 * lnh_ragged_color_input_rays: the same rows written ray by ray from the marcher's rays [N,3] (id, offset, count): a
 *   ray's samples share its direction (raymarching.cu:331-534), so the direction terms are evaluated once per ray from
 *   dirs[offset]; rows no ray owns (deltas[m,0] == 0: the unused tail, the slots of a ray dropped for lack of room) are
 *   set to zero.  Same values as lnh_ragged_color_input on the rows rays own.
 * lnh_ragged_color_output: rgb [M,2] f32 = sigmoid(y16[m, 0:2]) (network.py:224).  _backward: grad_y16 [M,16] =
 *   (grad_rgb * rgb * (1 - rgb) | 0).
 * lnh_ragged_grad_rows: grad_h16[m,0] = grad_sigma[m] * density_scale * exp(clamp(h16[m,0], -15, 15)) (trunc_exp backward,
 *   activation.py:17-19), grad_h16[m,1:16] = grad_cin[m, n_dir : n_dir + 15], n_dir = 3 + 6 * degree.
 */
LNH_API int lnh_ragged_points(const float *xyz, float bound, uint32_t M, float *x01, lnh_stream_t stream);
LNH_API int lnh_ragged_pack_weights(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                                    uint32_t ld_c0, uint32_t n_in, const float *wc1, uint32_t ld_c1, const float *wc2,
                                    uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream);
LNH_API int lnh_ragged_color_input(const float *dirs, const void *h16, uint32_t M, uint32_t degree, void *cin,
                                   lnh_stream_t stream);
LNH_API int lnh_ragged_color_input_rays(const float *dirs, const void *h16, const int32_t *rays, const float *deltas,
                                        uint32_t N, uint32_t M, uint32_t degree, void *cin, lnh_stream_t stream);
LNH_API int lnh_ragged_color_output(const void *y16, uint32_t M, float *rgb, lnh_stream_t stream);
LNH_API int lnh_ragged_color_output_backward(const float *grad_rgb, const float *rgb, uint32_t M, void *grad_y16,
                                             lnh_stream_t stream);
LNH_API int lnh_ragged_grad_rows(const float *grad_sigma, float density_scale, const void *h16, const void *grad_cin,
                                 uint32_t degree, uint32_t M, void *grad_h16, lnh_stream_t stream);
/*
 * lnh_lidar_color_forward: LiDAR colour head (network.py:199-237 with cal_lidar_color=True) on merged samples:
 * rgb[n,i,0:2] = sigmoid(MLP([freq(d_n) | geo_feat(sample)])) where weights[n,i] > 1e-4, else 0.
 * h16 [N*T,16] sigma-net rows in point order, perm [N,T], cdir [N,64] f32 = W0[:, :75] freq(d_n) (per ray),
 * w16 flat fp16: W0g [64,16] (col 0 zero, cols 1..15 = W0[:,75:90]) | W1 [64,64] | W2 padded to [16,64].
 */
LNH_API int lnh_lidar_color_forward(const void *h16, const int32_t *perm, const float *weights, const float *cdir,
                                    const void *w16, uint32_t N, uint32_t T, float *rgb, lnh_stream_t stream);
/*
 * lnh_lidar_color_composite_forward: the forward tail of the fused render step in one launch, one wave per ray —
 * lnh_lidar_merge_weights (renderer.py:217-243) + lnh_lidar_color_forward + lnh_lidar_composite_forward
 * (renderer.py:233-271) with the weights and the merge permutation of a ray kept in LDS in between.  Inputs as for those
 * three (z [N,T] merged depths, sigma_pt [N,T] point order, perm from lnh_lidar_resample); outputs sigma_m, weights
 * [N,T], rgb [N,T,2], weights_sum, depth [N], image [N,2].  sigma_m, weights, weights_sum and depth are bit-identical
 * to the separate entry points, image to fp32 rounding (different summation order).  T <= 2048.
 */
LNH_API int lnh_lidar_color_composite_forward(const float *z, const float *sigma_pt, const int32_t *perm,
                                              const float *sample_dist, const void *h16, const float *cdir,
                                              const void *w16, uint32_t N, uint32_t T, float density_scale,
                                              float *sigma_m, float *weights, float *rgb, float *weights_sum,
                                              float *depth, float *image, lnh_stream_t stream);
/*
 * lnh_lidar_color_backward: grad_rgb [N,T,2], grad_sigma [N,T] (merged order, from lnh_lidar_composite_backward)
 * -> grad_h16 [N*T,16] fp16 in POINT order (col 0 = grad_sigma * exp(clamp(pre,-15,15)), activation.py:17-19;
 * cols 1..15 = colour-head input gradient), grad_w fp32 flat like w16 (+=, fixed-order sum: wgrad_ws), ray_sum [N,64] f32 = sum over
 * the ray of d(hidden0) (multiply by freq(d) to get the gradient of W0[:, :75]).
 */
LNH_API int lnh_lidar_color_backward(const float *grad_rgb, const float *grad_sigma, const void *h16,
                                     const int32_t *perm, const float *weights, const float *cdir, const void *w16,
                                     uint32_t N, uint32_t T, void *grad_h16, float *grad_w, float *ray_sum,
                                     void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
/* The same with grad_rgb formed on the fly from grad_image [N,2] = d loss / d image: grad_rgb[n,i,:] = weights[n,i] *
 * grad_image[n,:], which is all lnh_lidar_composite_backward would have written there (call it with grad_rgb = NULL):
 * the [N,T,2] gradient never travels through HBM. */
LNH_API int lnh_lidar_color_backward_image(const float *grad_image, const float *grad_sigma, const void *h16,
                                           const int32_t *perm, const float *weights, const float *cdir,
                                           const void *w16, uint32_t N, uint32_t T, void *grad_h16, float *grad_w,
                                           float *ray_sum, void *wgrad_ws, uint64_t wgrad_ws_bytes,
                                           lnh_stream_t stream);

/* ---- range image <-> point cloud (lidarnerf/convert.py:99-160, 194-237) ---------------------------
 * lnh_lidar_to_pano: points [N,4] f32 (x,y,z,intensity) in the sensor frame -> pano [H,W] f32 (distance of the nearest
 *   point per pixel, 0 = empty) and intensities [H,W] f32; lidar_K = (fov_up, fov) in degrees; points with
 *   dist >= max_depth or outside the image are dropped; ties keep the earlier point.  keys_scratch: H*W*8 bytes.
 * lnh_pano_to_lidar: pano (+ optional intensities) -> points [H*W,4] and valid [H*W] u8 (pano != 0); the caller
 *   compacts in pixel order.
 */
LNH_API int lnh_lidar_to_pano(const float *points, uint32_t N, uint32_t H, uint32_t W, float fov_up, float fov,
                              float max_depth, void *keys_scratch, float *pano, float *intensities,
                              lnh_stream_t stream);
LNH_API int lnh_pano_to_lidar(const float *pano, const float *intensities, uint32_t H, uint32_t W, float fov_up,
                              float fov, float *points, uint8_t *valid, lnh_stream_t stream);

/* ---- evaluation: nearest-neighbour pass of the chamfer distance (extern/chamfer3D/chamfer3D.cu:9-138)
 * dist[j] = min_k |xyz1[j] - xyz2[k]|^2 (squared), idx[j] = the first k attaining it; xyz* are [n,3] / [m,3] f32.
 */
LNH_API int lnh_chamfer_nn(const float *xyz1, uint32_t n, const float *xyz2, uint32_t m, float *dist, int32_t *idx,
                           lnh_stream_t stream);

/* ---- optimizer step of the hash table (nerf/utils.py:1206-1226: GradScaler + torch.optim.Adam, fused) ------------
 * lnh_grad_check_f16: *found_inf = 1 if any of the n fp16 gradient values is inf / nan (never clears it).
 * lnh_adam_table_step: torch.optim.Adam (no weight decay / amsgrad) on fp32 param / exp_avg / exp_avg_sq with the
 *   fp16 gradient scaled by *inv_scale; also writes the fp16 copy of the updated parameters.  *found_inf != 0 skips
 *   the whole update.  The step counter t lives on the device, double-buffered: reads *step_in, writes *step_out.
 * lnh_adam_table_step_dlr: the same with the learning rate read from device memory (*lr, f32) — for a training step
 *   captured in a hipGraph, whose kernel arguments are frozen at capture while the schedule moves lr every step
 *   (main_lidarnerf.py:408-410: LambdaLR).
 */
LNH_API int lnh_grad_check_f16(const void *grad16, uint64_t n, float *found_inf, lnh_stream_t stream);
LNH_API int lnh_adam_table_step(float *param, float *exp_avg, float *exp_avg_sq, const void *grad16, void *param16,
                                uint64_t n, double lr, double beta1, double beta2, double eps,
                                const float *inv_scale, const float *found_inf, const float *step_in,
                                float *step_out, lnh_stream_t stream);
LNH_API int lnh_adam_table_step_dlr(float *param, float *exp_avg, float *exp_avg_sq, const void *grad16, void *param16,
                                    uint64_t n, const float *lr, double beta1, double beta2, double eps,
                                    const float *inv_scale, const float *found_inf, const float *step_in,
                                    float *step_out, lnh_stream_t stream);

/* ---- the whole optimizer step of a training iteration as two launches (nerf/utils.py:1216-1226: scaler.step(optimizer),
 * scaler.update(), lr_scheduler.step(); main_lidarnerf.py:389-391, 408-410: Adam(betas .9/.99, eps 1e-15), lr0 * 0.1^(it/iters))
 * The scalars of the step live in ONE device buffer `state` of LNH_TRAIN_STATE_FLOATS floats (indices LNH_TS_*), so a
 * training step captured in a hipGraph needs no host value:
 *   SCALE / GROWTH   GradScaler's loss scale and growth counter          T / T_NEXT     Adam's step count, before / after
 *   FOUND            stamp IT + 1 of the last step that saw an inf/nan   IT / IT_NEXT   scheduler steps taken, before / after
 *   INV / INV_TABLE  1 / (scale * div_small), 1 / (scale * div_table)    LR             lr0 * 0.1^min(IT / iters, 1)
 *   LAST_SCALE       the scale the gradients of the last step carry      SKIPPED        1 if the last step was skipped
 * lnh_train_check: commits T_NEXT / IT_NEXT of the previous step, forms INV / INV_TABLE / LAST_SCALE / LR, and stamps
 *   FOUND = IT + 1 if the n16 fp16 values at grad16 or any of the small fp32 gradients holds an inf / nan (idempotent:
 *   may be called once per piece of a gradient that arrives in pieces; a MAX all-reduce of FOUND over ranks keeps the
 *   stamp).  div_table / div_small = what the SUMMED gradients still have to be divided by (data parallel: the world size).
 * lnh_train_step: Adam with torch's fused arithmetic on the fp32 table (n values, fp16 gradient, also writes the fp16
 *   copy; n = 0: the table is stepped elsewhere, e.g. lnh_adam_table_step_dlr per shard with lr = &state[LNH_TS_LR],
 *   inv_scale = &state[LNH_TS_INV_TABLE], found_inf = &state[LNH_TS_SKIPPED], step_in / step_out = &state[LNH_TS_T] /
 *   &state[LNH_TS_T_NEXT]) and on n_small <= LNH_TRAIN_MAX_SMALL fp32 tensors (host arrays of device pointers; a null
 *   gradient skips that tensor; their moments lie back to back in small_exp_avg / small_exp_avg_sq in the order given),
 *   skipped as a whole when FOUND == IT + 1; then T_NEXT, IT_NEXT, SKIPPED and torch's amp_update_scale_.
 * lnh_zero_regions: clears up to 8 device regions (host arrays of pointers and byte counts, 4-byte granular) with ONE launch.
 */
#define LNH_TRAIN_STATE_FLOATS 16
#define LNH_TRAIN_MAX_SMALL 16
#define LNH_TS_SCALE 0
#define LNH_TS_GROWTH 1
#define LNH_TS_FOUND 2
#define LNH_TS_INV 3
#define LNH_TS_INV_TABLE 4
#define LNH_TS_LAST_SCALE 5
#define LNH_TS_T 6
#define LNH_TS_IT 7
#define LNH_TS_LR 8
#define LNH_TS_T_NEXT 9
#define LNH_TS_IT_NEXT 10
#define LNH_TS_SKIPPED 11
LNH_API int lnh_train_check(float *state, const void *grad16, uint64_t n16, const float *const *small_grads,
                            const uint32_t *small_numel, uint32_t n_small, float div_table, float div_small, double lr0,
                            double iters, lnh_stream_t stream);
LNH_API int lnh_train_step(float *state, float *param, float *exp_avg, float *exp_avg_sq, const void *grad16,
                           void *param16, uint64_t n, float *const *small_params, const float *const *small_grads,
                           const uint32_t *small_numel, uint32_t n_small, float *small_exp_avg, float *small_exp_avg_sq,
                           double beta1, double beta2, double eps, double growth_factor, double backoff_factor,
                           uint32_t growth_interval, lnh_stream_t stream);
LNH_API int lnh_zero_regions(void *const *ptrs, const uint64_t *bytes, uint32_t count, lnh_stream_t stream);


/* ------------------------------------------------------------------ bf16 MLP operands (BASELINE config 5) ---- */
/*
 * The same eleven entry points with v_mfma_f32_16x16x32_bf16 operands ("fp16 hash features + bf16 MFMA MLP"; the
 * reference reaches the MLPs through torch.autocast, lidarnerf/nerf/utils.py:626,1212 — under
 * autocast(dtype=torch.bfloat16) its Linear stacks run in bf16 while the grid encoder keeps casting its table to half,
 * gridencoder/grid.py:54-57).  Every buffer that holds MLP-side 16-bit data is bf16 here — inputs / outputs /
 * forward_buffer / grad / grad_inputs of lnh_mlp_*_bf16, the packed weights, h16 and grad_h16 — while the hash-grid
 * `features` entering lnh_density_mlp_forward_bf16 and the `grad_features` leaving lnh_density_mlp_backward_bf16 stay
 * fp16 (the grid kernels' type).  Accumulation is fp32, weight gradients are fp32, as in the fp16 build.
 */
LNH_API int lnh_mlp_forward_bf16(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim,
                            uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats, uint32_t activation,
                            uint32_t output_activation, void *forward_buffer, void *outputs, lnh_stream_t stream);
LNH_API int lnh_mlp_backward_bf16(const void *grad, const void *inputs, const void *weights, uint32_t B,
                             uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats,
                             uint32_t activation, uint32_t output_activation, void *grad_inputs, float *grad_weights,
                             void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
LNH_API int lnh_mlp_backward_data_bf16(const void *grad, const void *forward_buffer, const void *weights_t, uint32_t B,
                                       uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                       uint32_t n_hidden_mats, uint32_t activation, void *backward_buffer,
                                       void *grad_inputs, lnh_stream_t stream);
LNH_API int lnh_mlp_wgrad_bf16(const void *grad, const void *acts, uint32_t B, uint32_t M, uint32_t N, float *grad_weights,
                               void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
LNH_API int lnh_density_mlp_forward_bf16(const void *features, const void *weights, uint32_t B, uint32_t T_cur,
                                    uint32_t T_tot, uint32_t slot_off, uint32_t feat_rows, void *h16, float *sigma,
                                    lnh_stream_t stream);
LNH_API int lnh_density_mlp_backward_bf16(const void *grad_h16, const void *features, const void *weights, uint32_t B,
                                     uint32_t T_cur, uint32_t T_tot, uint32_t slot_off, void *grad_features,
                                     float *grad_weights, void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
LNH_API int lnh_ragged_pack_weights_bf16(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                                         uint32_t ld_c0, uint32_t n_in, const float *wc1, uint32_t ld_c1, const float *wc2,
                                         uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream);
LNH_API int lnh_ragged_color_input_bf16(const float *dirs, const void *h16, uint32_t M, uint32_t degree, void *cin,
                                        lnh_stream_t stream);
LNH_API int lnh_ragged_color_input_rays_bf16(const float *dirs, const void *h16, const int32_t *rays, const float *deltas,
                                             uint32_t N, uint32_t M, uint32_t degree, void *cin, lnh_stream_t stream);
LNH_API int lnh_ragged_color_output_bf16(const void *y16, uint32_t M, float *rgb, lnh_stream_t stream);
LNH_API int lnh_ragged_color_output_backward_bf16(const float *grad_rgb, const float *rgb, uint32_t M, void *grad_y16,
                                                  lnh_stream_t stream);
LNH_API int lnh_ragged_grad_rows_bf16(const float *grad_sigma, float density_scale, const void *h16, const void *grad_cin,
                                      uint32_t degree, uint32_t M, void *grad_h16, lnh_stream_t stream);
LNH_API int lnh_lidar_dir_term_bf16(const float *dir_features, const float *w0, uint32_t ldw, uint32_t N, uint32_t K,
                               float *features16, float *cdir, lnh_stream_t stream);
LNH_API int lnh_lidar_dir_term_freq_bf16(const float *dirs, uint32_t degree, const float *w0, uint32_t ldw, uint32_t N,
                                         float *features16, float *cdir, lnh_stream_t stream);
LNH_API int lnh_lidar_pack_weights_bf16(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1,
                                   const float *wc0, uint32_t ld_c0, uint32_t n_dir, const float *wc1, uint32_t ld_c1,
                                   const float *wc2, uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream);
LNH_API int lnh_lidar_step_prologue_bf16(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                                         uint32_t ld_c0, uint32_t degree, const float *wc1, uint32_t ld_c1, const float *wc2,
                                         uint32_t ld_c2, void *wsig16, void *wcol16, const float *u, const float *rays_o,
                                         const float *rays_d, const float *aabb, float bound, uint32_t N, uint32_t T,
                                         uint32_t T_tot, float near, float far, float *z, float *x01, float *features16,
                                         float *cdir, lnh_stream_t stream);
LNH_API int lnh_ragged_color_forward_bf16(const void *h16, const int32_t *rays, const float *cdir, const void *w16, uint32_t N,
                                          uint32_t M, float *rgb, lnh_stream_t stream);
LNH_API int lnh_ragged_color_backward_bf16(const float *grad_rgb, const float *grad_sigma, float density_scale,
                                           const void *h16, const int32_t *rays, const float *cdir, const void *w16,
                                           uint32_t N, uint32_t M, void *grad_h16, float *grad_w, float *ray_sum,
                                           void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
LNH_API int lnh_lidar_color_forward_bf16(const void *h16, const int32_t *perm, const float *weights, const float *cdir,
                                    const void *w16, uint32_t N, uint32_t T, float *rgb, lnh_stream_t stream);
LNH_API int lnh_lidar_color_composite_forward_bf16(const float *z, const float *sigma_pt, const int32_t *perm,
                                                   const float *sample_dist, const void *h16, const float *cdir,
                                                   const void *w16, uint32_t N, uint32_t T, float density_scale,
                                                   float *sigma_m, float *weights, float *rgb, float *weights_sum,
                                                   float *depth, float *image, lnh_stream_t stream);
LNH_API int lnh_lidar_color_backward_bf16(const float *grad_rgb, const float *grad_sigma, const void *h16,
                                     const int32_t *perm, const float *weights, const float *cdir, const void *w16,
                                     uint32_t N, uint32_t T, void *grad_h16, float *grad_w, float *ray_sum,
                                     void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream);
LNH_API int lnh_lidar_color_backward_image_bf16(const float *grad_image, const float *grad_sigma, const void *h16,
                                                const int32_t *perm, const float *weights, const float *cdir,
                                                const void *w16, uint32_t N, uint32_t T, void *grad_h16,
                                                float *grad_w, float *ray_sum, void *wgrad_ws,
                                                uint64_t wgrad_ws_bytes, lnh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LIDARNERF_HIP_H */
