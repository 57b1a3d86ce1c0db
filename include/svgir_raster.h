/* include/svgir_raster.h -- C ABI of libsvgir_raster.so (MI355X / gfx950 surfel rasterizer).
 *
 * This is the drop-in boundary for the hot path of learner-shx/SVG-IR: it replaces the two
 * `CudaRasterizer::Rasterizer` classes that the reference's torch glue binds
 *     svgss_rasterization/cuda_rasterizer/rasterizer.h:24-115   (forward / backward / markVisible, "svgss")
 *     rgss-rasterization/cuda_rasterizer/rasterizer.h:24-104    (same, "rgss")
 * and that `svgss_rasterization/rasterize_points.cu:35-286` / `rgss-rasterization/rasterize_points.cu:36-264`
 * call with raw device pointers.  No torch, HIP or C++ types appear in the signatures: plain pointers, sizes and
 * a `void*` stream (a hipStream_t).  All pointers are DEVICE pointers unless stated otherwise.  All arithmetic
 * is fp32; images are CHW row-major; per-Gaussian tensors are row-major [P,k] exactly as the reference's.
 *
 * Errors: every entry point returns a negative svgir_status on failure and sets a thread-local message
 * retrievable with svgir_last_error() (the reference throws std::runtime_error / AT_ERROR instead,
 * rasterize_points.cu:65-67, rasterizer_impl.cu:264-267; the host bindings convert the code to RuntimeError).
 */
#ifndef SVGIR_RASTER_H_
#define SVGIR_RASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVGIR_ABI_VERSION 14

enum svgir_variant { SVGIR_RGSS = 0, SVGIR_SVGSS = 1 };

enum svgir_status {
    SVGIR_OK = 0,
    SVGIR_ERR_INVALID = -1,  /* bad shapes / missing inputs / unsupported channel counts */
    SVGIR_ERR_HIP = -2,      /* a HIP runtime call or kernel launch failed */
    SVGIR_ERR_ALLOC = -3     /* a blob allocation callback returned NULL */
};

/* Blob allocator: must return a device pointer to at least `bytes` bytes (256-byte aligned), valid until the
 * matching backward has run.  Replaces the reference's std::function<char*(size_t)> resize lambdas
 * (rasterize_points.cu:27-33). */
typedef char* (*svgir_alloc_fn)(size_t bytes, void* ctx);

typedef struct svgir_fused_shade svgir_fused_shade;   /* below, behind the shading entry points */

/* Inputs of one view.  Field-for-field the arguments of Rasterizer::forward
 * (svgss rasterizer_impl.cu:209-242, rgss :209-241). */
typedef struct svgir_params {
    int32_t variant;            /* svgir_variant */
    int32_t P;                  /* #Gaussians */
    int32_t S;                  /* #feature channels (svgss <= 50, rgss <= 33; Q9) */
    int32_t VS;                 /* #vfeature channels, multiple of 4, VS/4 <= 20 (svgss only) */
    int32_t D;                  /* active SH degree 0..3 */
    int32_t M;                  /* SH coefficients per colour channel in `shs` */
    int32_t W, H;               /* image size */
    const float* background;    /* [3] */
    const float* means3D;       /* [P,3] */
    const float* shs;           /* [P,M,3] or NULL */
    const float* colors_precomp;/* [P,3] or NULL (exactly one of shs / colors_precomp) */
    const float* features;      /* [P,S] or NULL when S == 0 */
    const float* vfeatures;     /* [P,VS] or NULL when VS == 0 */
    const float* opacities;     /* [P] */
    const float* scales;        /* [P,3] or NULL */
    const float* rotations;     /* [P,4] (r,x,y,z), NOT normalised in-kernel (Q3), or NULL */
    const float* cov3D_precomp; /* [P,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
    const float* viewmatrix;    /* [16] = W2C transposed (column-major W2C) */
    const float* projmatrix;    /* [16] = full projection, same convention */
    const float* cam_pos;       /* [3] */
    const float* prcppoint;     /* [2]  svgss; carried for API parity, unused by the arithmetic (Q11) */
    const float* patchbbox;     /* [4]  svgss: h0,w0,h1,w1 in pixels */
    const float* config;        /* [config_len] floats ON THE DEVICE, read by the kernels like the reference does;
                                   svgss: surface, normalize_depth, per_pixel_depth, (lrn_cam).  Entries >=
                                   config_len read as 0 (Q7: the reference reads config[3] out of bounds).
                                   rgss ignores it (compile-time {1,1,1} in the reference). */
    int32_t config_len;
    float scale_modifier;
    float tan_fovx, tan_fovy;
    float cx, cy;               /* rgss: principal point for surface_xyz */
    int32_t prefiltered;
    int32_t computer_pseudo_normal; /* rgss */
    int32_t backward_geometry;      /* rgss */
    int32_t debug;                  /* synchronise + check after every kernel (reference CHECK_CUDA) */
    void* features_ready;           /* optional (NULL: `features` / `vfeatures` are complete on `stream` at call time): a HIP
                                       event (hipEvent_t) recorded on ANOTHER stream that is still producing them.  Only the
                                       composite kernel of svgir_forward waits for it -- everything in front of the composite
                                       (preprocess, sorts, emit, ranges, cull) neither reads the features nor waits -- which
                                       is how the per-splat shading (svgir_shade_forward on a side stream) overlaps the
                                       binning of the same view.  Must stay alive until svgir_forward returns; ignored by
                                       svgir_backward.  (ABI 9: replaces the thread-local svgir_forward_wait_features of
                                       ABI 8 -- no state survives between calls.) */
    int32_t forward_only;           /* != 0 (ABI 12): no backward will follow (evaluation loops).  The composite keeps no blend states for
                                       the backward's depth segments -- nothing is dumped, the binning blob holds no state slots.  A
                                       svgir_backward on such a forward still works: it replays the composite once to produce them. */
    const svgir_fused_shade* shade; /* optional (svgss, ABI 12): the per-splat shading of this view, run INSIDE svgir_forward /
                                       svgir_backward for the surfels the view's composite actually reads (see svgir_fused_shade);
                                       `features` / `vfeatures` are then OUTPUT buffers of svgir_forward.  NULL: the caller shaded. */
    int32_t workload_scope;         /* (ABI 14) scope of the speculation history, 0 = the process-wide default.  svgir_forward sizes its
                                       speculative launches (instance capacity, state slots, three-pass depth sort) from the recent views of
                                       the same WORKLOAD = (device, W, H, S, VS, variant, workload_scope, Gaussian count within a factor of
                                       two).  Two models that share all of that and alternate in one process would feed one history: give
                                       each its own scope id.  Results never depend on it -- only how often a view's dependent stages are
                                       re-run.  See svgir_reset_workload_history. */
} svgir_params;

/* Outputs of forward.  Every buffer is written completely: the caller need not clear any of them (the reference's glue
 * hands zero-filled tensors to its kernels, rasterize_points.cu:76-88; here the kernels that run anyway write the zeros). */
typedef struct svgir_outputs {
    float* out_color;         /* [3,H,W] */
    float* out_normal;        /* [3,H,W] */
    float* out_depth;         /* [1,H,W] */
    float* out_opacity;       /* [1,H,W] */
    float* out_feature;       /* [S,H,W] */
    float* out_vfeature;      /* [VS/4,H,W]  svgss */
    float* out_pseudo_normal; /* [3,H,W]     rgss; zero without computer_pseudo_normal and at pixels with a
                                 degenerate stencil (forward.cu:620-622) */
    float* out_surface_xyz;   /* [3,H,W]     rgss; zero without computer_pseudo_normal */
    float* out_weights;       /* [P] */
    int32_t* radii;           /* [P] */
} svgir_outputs;

/* Upstream gradients and gradient outputs of backward (Rasterizer::backward, svgss rasterizer_impl.cu:386-432,
 * rgss :411-449).  The caller need not clear the dL_d* outputs (the reference's glue zero-fills them,
 * rasterize_points.cu:195-211): svgir_backward clears them itself before the per-Gaussian kernels write the gradients of
 * the visible Gaussians.  A caller that lays all outputs out in one allocation (16-byte aligned, a multiple of 16 bytes) passes it as
 * clear_base / clear_bytes: the waves of the composite backward -- which accumulates in the scratch, not in these tensors -- then zero
 * it in passing, no memset and no second stream; without the hint the tensors are cleared one by one on an internal side stream that is
 * joined behind the composite backward. */
typedef struct svgir_grads {
    /* upstream gradients: any of the six may be NULL = all zero (an output that took no part in the loss: the training loss of the
     * reference reads one image, svgss.py:281-289) -- nothing is read for it, and the binder need not allocate and fill a zero image */
    const float* dL_dout_color;    /* [3,H,W] */
    const float* dL_dout_normal;   /* [3,H,W] */
    const float* dL_dout_depth;    /* [1,H,W] */
    const float* dL_dout_opacity;  /* [1,H,W] */
    const float* dL_dout_feature;  /* [S,H,W] */
    const float* dL_dout_vfeature; /* [VS/4,H,W] svgss */
    float* dL_dmeans2D;   /* [P,3] (x,y used) */
    float* dL_dconic;     /* [P,4] (x,y,w used) */
    float* dL_dopacity;   /* [P] */
    float* dL_dcolors;    /* [P,3] */
    float* dL_dfeatures;  /* [P,S] */
    float* dL_dvfeatures; /* [P,VS] svgss */
    float* dL_dnormal;    /* [P,3] */
    float* dL_ddepth;     /* [P] */
    float* dL_dmeans3D;   /* [P,3] */
    float* dL_dcov3D;     /* [P,6] */
    float* dL_dsh;        /* [P,M,3] */
    float* dL_dscales;    /* [P,3] */
    float* dL_drotations; /* [P,4] */
    float* dL_dviewmat;   /* [16] svgss (zero unless config[3] > 0) */
    float* dL_dprojmat;   /* [16] svgss */
    float* dL_dcampos;    /* [3]  svgss */
    void* clear_base;     /* optional: a region that contains every dL_d* output above and nothing the library must */
    size_t clear_bytes;   /*           preserve (NULL / 0: the outputs are cleared one by one) */
    /* Fused shading (svgir_params.shade != NULL; ABI 12): gradients of the shading inputs, layouts as svgir_shade_backward's outputs,
     * every one written completely (rows of surfels that received no blend weight in this view are zero).  The four per-surfel tensors
     * MAY lie inside clear_base (all four or none): their zero rows then come from the composite backward's clearing sweep instead of a
     * zero-fill launch; dL_denv and env_grad_work must lie outside. */
    float* dL_dbase_color;      /* [P,12] */
    float* dL_droughness;       /* [P,4] */
    float* dL_dshade_normals;   /* [P,4,3] */
    float* dL_dradiance;        /* [P,Ns,3]; may be NULL when shade->sp.radiance_ratio is set (ABI 13: the cache is detached) */
    float* dL_denv;             /* [env_h,env_w,3] */
    float* env_grad_work;       /* scratch, env_h*env_w*3 (+ SVGIR_SHADE_RATIO_WORK with dL_dradiance_ratio) floats */
    const float* dL_dreduced;   /* optional [P,70]: upstream gradient of svgir_fused_shade.reduced (needs all_surfels != 0) */
    const float* out_weights;   /* optional (required by the fused shading): the forward's out_weights [P].  A surfel that received no
                                 * blend weight has no gradient rows and all-zero composite gradients; with the weights at hand the
                                 * per-Gaussian kernels behind the composite (row reduction, cov2D / preprocess / SH backward, the fused
                                 * shading's backward) walk the list of blended surfels -- 13-29 % of the model on the BASELINE scenes --
                                 * instead of all P.  Used where building the list costs less than it saves: svgss with vfeatures from 50 000
                                 * surfels on, otherwise from 400 000. */
    float* dL_dradiance_ratio;  /* optional [1] (ABI 13; needs shade->sp.radiance_ratio): see svgir_shade_backward */
} svgir_grads;

int svgir_abi_version(void);

/* Sizes of the three opaque scratch blobs (the reference's required<GeometryState/ImageState/BinningState>,
 * rasterizer_impl.h:73-84). */
size_t svgir_geom_bytes(int32_t P);
size_t svgir_image_bytes(int32_t W, int32_t H);
/* The binning blob also holds the per-segment forward states that parallelise the backward over depth, hence the
 * dependence on the image size and channel counts (S features, VS vfeature floats; VS = 0 for rgss). */
size_t svgir_binning_bytes(int32_t num_rendered, int32_t W, int32_t H, int32_t S, int32_t VS);   /* worst case: the forward asks its
 * allocator for less once it has seen a view of the workload (state slots from the pair statistics of recent views; such blobs are
 * an odd multiple of 128 bytes long and belong to the forward that laid them out: pass them to svgir_backward unchanged) */
/* Byte offset of the int32 n_contrib[H*W] plane inside the image blob (rgss returns a view of it, Q10). */
size_t svgir_image_ncontrib_offset(int32_t W, int32_t H);

/* Forward pass.  Replaces CudaRasterizer::Rasterizer::forward (svgss rasterizer_impl.cu:209-382,
 * rgss :209-407).  Calls geom(), image() and binning().  From the second call on, binning() is called -- and the
 * count-dependent stages (emit, tile sort, ranges, composite) are launched -- speculatively for a capacity derived
 * from the previous call's instance count, while the GPU still computes the count; the stages read the count on the
 * device.  The count reaches the host as a tagged store into pinned memory (no copy, no event) and then only confirms the guess (no
 * GPU idle time); if the guess was too small, binning() is called again and those stages are re-run (only the LAST pointer it returned is
 * used).
 * Returns num_rendered (R >= 0) or a negative svgir_status. */
int svgir_forward(const svgir_params* p, const svgir_outputs* o,
                  svgir_alloc_fn geom, void* geom_ctx,
                  svgir_alloc_fn binning, void* binning_ctx,
                  svgir_alloc_fn image, void* image_ctx,
                  void* stream);

/* Several views in flight from ONE host thread (ABI 12).  Every view is begun -- validated, its blobs allocated, ALL of its kernels
 * launched on its stream, the count-dependent ones speculatively -- before the first view's instance count is awaited.  Views on distinct
 * streams overlap on the GPU (a single 800 x 800 view leaves the SIMDs under-occupied: two views in flight render 1.3x the surfels per
 * second); the results are bit-identical to `count` svgir_forward calls.  num_rendered receives each view's R or its negative status;
 * the return value is 0 or the first error.  The reference has no counterpart (its forward blocks on a cudaMemcpy per view,
 * rasterizer_impl.cu:307-312; eval_relighting_tensoIR.py:303-378 renders its views one after the other). */
typedef struct svgir_view_call {
    const svgir_params* params; const svgir_outputs* outputs;
    svgir_alloc_fn geom; void* geom_ctx;
    svgir_alloc_fn binning; void* binning_ctx;
    svgir_alloc_fn image; void* image_ctx;
    void* stream;
    int32_t num_rendered;   /* out */
} svgir_view_call;
int svgir_forward_batch(svgir_view_call* views, int32_t count);

/* What the speculative launches of svgir_forward did so far in this process (monitoring / tests; ABI 11):
 * out5 = {forwards, views re-run because the instance capacity guessed from the previous views was too small, views whose blend states
 * were dumped again by their backward because the state-slot capacity was, views re-run because a visible depth key did not carry the
 * top byte the last views' keys shared (the depth sort then runs three 8-bit passes instead of four), views whose depth sort ran three
 * passes}.  The reference has no counterpart: its forward waits for the
 * instance count (rasterizer_impl.cu:307-312) and always sorts 64-bit keys. */
void svgir_speculation_stats(int64_t* out5);
/* Forgets the speculation history (ABI 14) of one scope (svgir_params.workload_scope), or of every scope (scope < 0): the next view of
 * such a workload is launched like a first view (exact instance capacity after a host wait, state slots sized from its own cull, four
 * depth passes).  For callers that recycle scope ids, or that switch scenes under one id.  Blobs of earlier forwards stay valid. */
void svgir_reset_workload_history(int32_t scope);

/* Backward pass.  Replaces CudaRasterizer::Rasterizer::backward (svgss rasterizer_impl.cu:386-523,
 * rgss :411-535).  `R` and the three blobs are what the matching svgir_forward produced; `radii` is its
 * radii output.  `binning_bytes` is the size the binning callback was last asked for: the blob is laid out for an
 * instance capacity >= R that the backward recovers from it (the forward sizes the blob before it knows R). */
int svgir_backward(const svgir_params* p, const svgir_grads* g, int32_t R, const int32_t* radii,
                   char* geom_blob, char* binning_blob, size_t binning_bytes, char* image_blob,
                   char* scratch, size_t scratch_bytes, void* stream);
/* Size of the backward scratch (device memory, contents irrelevant on entry, only needed during the call; REQUIRED).
 * The composite backward reduces every per-Gaussian gradient over the 64 pixels of a wave and then
 *   svgss: stores one complete gradient row per (instance, sub-tile) pair, summed per Gaussian in a fixed order by a
 *          second kernel (no atomics, bit-reproducible gradients): a 4-byte reverse-map entry per possible pair (4 * capacity)
 *          + the rows (worst case 4 * capacity; see svgir_backward_scratch_bytes_for);
 *   rgss : accumulates with float atomics into ONE packed row per Gaussian (P rows), unpacked into the dL_d* tensors
 *          by the per-Gaussian backward kernel.
 * The reference accumulates with one global float atomic per (pixel, splat, output) straight into its dL_d* tensors
 * (backward.cu:880-930) and needs no scratch; a binder allocates this buffer next to them. */
size_t svgir_backward_scratch_bytes(int32_t variant, int32_t P, size_t binning_bytes, int32_t W, int32_t H, int32_t S,
                                    int32_t VS);
/* The same for ONE view: `image_blob` is the image blob of the svgir_forward whose backward is about to run.  svgss with vfeatures
 * then gets one gradient row per (sub-tile, instance) pair that survived that view's cull (1.2 per instance on the BASELINE scenes)
 * instead of four per instance: the forward reads the pair count back asynchronously behind its cull, and this call returns at once
 * unless that count is still on its way (then it waits: polling, and beyond 50 ms blocking on the device).  A blob the library has not
 * seen in its last 1024 forwards is read itself (the image blob carries the view's counts and capacities); NULL gets the
 * worst case; svgir_backward accepts either size for the view it belongs to.  (ABI 10) */
size_t svgir_backward_scratch_bytes_for(int32_t variant, int32_t P, size_t binning_bytes, const char* image_blob, int32_t W, int32_t H,
                                        int32_t S, int32_t VS);

/* Introspection of the state blobs for tests / debugging (the blobs stay opaque to the rendering path): byte offset of
 * the depth-sorted instance list `point_list` (uint32 Gaussian ids, R entries; BinningState::point_list,
 * rasterizer_impl.h:66-76) inside the binning blob, and of the per-tile `ranges` (uint2 per tile; ImageState::ranges,
 * rasterizer_impl.h:49-63) inside the image blob. */
size_t svgir_binning_point_list_offset(size_t binning_bytes, const char* image_blob, int32_t W, int32_t H, int32_t S, int32_t VS);
size_t svgir_image_ranges_offset(int32_t W, int32_t H);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer_impl.cu:141-153).  `present` is a byte per
 * Gaussian.  svgss: the reference kernel body is commented out, so `present` is left untouched (all false, Q14);
 * rgss: present = view-space z > 0.2. */
int svgir_mark_visible(int32_t variant, int32_t P, const float* means3D, const float* viewmatrix,
                       const float* projmatrix, uint8_t* present, void* stream);

/* ---- per-splat spatially-varying BRDF shading ---------------------------------------------
 * Fused replacement of the PyTorch `rendering_equation4` + `GGX_specular4` (gaussian_renderer/svgss.py:537-631) with
 * the lat-long env lookup of `DirectLightMap.direct_light` / `EnvLight.direct_light`
 * (scene/direct_light_map.py:70-83, scene/envmap.py:53-72) and the feature packing of svgss.py:143-166.  There is no
 * reference C/CUDA counterpart (the reference does this with ~40 broadcasting torch ops and [P,Ns,4,3] temporaries).
 *
 * Layouts as in the reference: base_color [P,12] (channel*4 + corner), roughness [P,4], normals [P,4,3] (world
 * space), viewdirs [P,3], radiance / incident_dirs [P,Ns,3], visibility / incident_areas [P,Ns,1],
 * env [env_h, env_w, 3].  The looked-up light is env_scale * bilinear(f(env)) with f = softplus when
 * env_softplus != 0 (DirectLightMap: softplus, scale 2) or identity (EnvLight: scale 1), clamped to [0, 64].
 * `env_work` is a caller-provided, 16-byte aligned scratch of env_h*env_w*4 floats (holds f(env) as one float4 per texel).
 *
 * reduced [P,70]: pbr[12] diffuse_light[12] specular[12] direct[12] indirect[12] mean_incident[3] mean_local[3]
 *                 mean_global[3] mean_visibility[1]   (means over the Ns samples, like `.mean(-2)` in svgss.py)
 * features / vfeatures (either may be NULL): the rasterizer inputs of svgss.py:143-166;
 *   training != 0: features [P,4]  = [mean_vis, mean_local]; vfeatures [P,52] = [pbr, base_color, normal_view,
 *                  roughness, diffuse_light];   training == 0: features [P,7] = [mean_incident, mean_local, mean_vis],
 *                  vfeatures [P,64] = [pbr, base_color, normal_view, roughness, direct, indirect];
 *   normal_view = normals @ viewmatrix[:3,:3], stored channel*4 + corner. */
typedef struct svgir_shade_params {
    int32_t P, Ns, env_h, env_w;
    int32_t env_softplus;
    int32_t training;
    float env_scale;
    const float* base_color;
    const float* roughness;
    const float* normals;
    const float* viewdirs;
    const float* radiance;
    const float* visibility;
    const float* incident_dirs;
    const float* incident_areas;
    const float* env;
    const float* viewmatrix;   /* [16], only for the packed vfeatures; may be NULL when vfeatures is NULL */
    float* env_work;
    const float* env_transform; /* [9] row-major 3x3 or NULL: the env lookup uses transform * dir (EnvLight.transform,
                                 * scene/envmap.py:57-60); every other term keeps the untransformed direction */
    /* Incident directions generated in the kernels instead of streamed: when incident_dirs is
     * NULL, sample i of surfel g is the reference's Fibonacci hemisphere lattice around lattice_normals[g]
     * (fibonacci_sphere_sampling + rotation_between_z, utils/graphics_utils.py:9-37, utils/sh_utils.py:36-68; what
     * sample_incident_rays stores in pc._incident_dirs, scene/gaussian_model.py:23-31), with the per-surfel random
     * azimuth offset lattice_offsets[g] of the training branch (NULL: evaluation lattice).  incident_areas may then be
     * NULL too (the lattice's areas are the constant 2 pi).  lattice_work: scratch of 4 * Ns floats. */
    const float* lattice_normals;   /* [P,3] unit */
    const float* lattice_offsets;   /* [P] radians or NULL */
    float* lattice_work;
    /* Shade a SUBSET of the P surfels (ABI 12; both NULL: all of them).  `subset` [P] is a permutation of 0..P-1 in device memory whose
     * first *subset_count entries (a device uint32) are the surfels to shade; the output rows of the others (reduced / features /
     * vfeatures; the backward's dL_dbase_color, dL_droughness, dL_dnormals, dL_dradiance) are ZERO-filled by the same call, so every
     * output is still written completely.  Rows of shaded surfels are bit-identical to an all-P call. */
    const uint32_t* subset;
    const uint32_t* subset_count;
    /* The reference's radiance cache enters the shading as  get_radiances = nan_to_num(_radiances.detach() * _radiance_ratio, nan=0)
     * (scene/gaussian_model.py:323-324): a [P,Ns,3] product and its clean-up per iteration, and in the backward a [P,Ns,3] gradient
     * whose only use is the sum that gives dL/d_radiance_ratio.  ABI 13: `radiance_ratio` (NULL: `radiance` is used as it is) points
     * to that scalar in device memory; the kernels then use nan_to_num(radiance * ratio) (torch.nan_to_num: NaN -> 0, +-inf -> the
     * largest finite floats) -- the same fp32 product, bit for bit -- and the backward can return the scalar's gradient directly. */
    const float* radiance_ratio;
} svgir_shade_params;

#define SVGIR_SHADE_REDUCED 70
#define SVGIR_SHADE_RATIO_WORK 256   /* extra floats of env_grad_work when dL_dradiance_ratio is requested */

int svgir_shade_forward(const svgir_shade_params* p, float* reduced, float* features, float* vfeatures, void* stream);

/* Backward of svgir_shade_forward w.r.t. base_color, roughness, normals, radiance and the env texels, given any of
 * dL/d(reduced) [P,70], dL/d(features) [P,S] and dL/d(vfeatures) [P,VS] (each may be NULL, not all three; the
 * packed rows follow p->training as in the forward and need p->viewmatrix; the gradient of mean_visibility is
 * ignored: visibility is not differentiable in the reference either).  The packing's direct terms (vfeatures
 * carries base_color, view-space normals and roughness, svgss.py:152-166) are included.  All outputs are
 * overwritten; dL_denv [env_h,env_w,3] is the gradient w.r.t. the RAW env (softplus' included).
 * `env_grad_work`: scratch of env_h*env_w*3 floats.
 * With p->radiance_ratio (ABI 13): dL_dradiance is the gradient w.r.t. the RAW radiance (ratio and the nan_to_num mask included) and
 * may be NULL (the reference detaches the cache); dL_dradiance_ratio [1] (optional, NULL without p->radiance_ratio) receives
 * sum(dL/d(incident) * isfinite(radiance * ratio) * radiance), what autograd returns for the scalar -- summed per workgroup and then
 * in a fixed order (reproducible); env_grad_work then holds SVGIR_SHADE_RATIO_WORK more floats. */
int svgir_shade_backward(const svgir_shade_params* p, const float* dL_dreduced, const float* dL_dfeatures,
                         const float* dL_dvfeatures, float* dL_dbase_color,
                         float* dL_droughness, float* dL_dnormals, float* dL_dradiance, float* dL_denv,
                         float* env_grad_work, float* dL_dradiance_ratio, void* stream);

/* Shading fused into the rasterizer calls (svgir_params.shade; ABI 12).  The only consumers of the shading's outputs are the packed
 * `features` / `vfeatures` rows the svgss composite blends (gaussian_renderer/svgss.py:143-182), and it reads the rows of the surfels
 * that survive the view's culls only -- 44 % of the surfels pass the preprocess culls on the BASELINE scenes, ~30 % survive the per-tile
 * cull as well, and 13-29 % ever receive a blend weight -- while the reference (and svgir_shade_forward on its own) shades all P.
 *   svgir_forward : runs preprocess / sorts / per-tile cull first, then shades exactly the surfels that are a candidate of at least one
 *       8x8 sub-tile -- or, when sp.Ns >= 128 (evaluation sample counts: shading a surfel costs far more than compositing it), the
 *       surfels a geometry-only pre-pass of the composite finds to receive a blend weight -- writes their `features` [P,S] / `vfeatures` [P,VS] rows (S, VS = 4, 52 when sp.training, else 7, 64) and zero-fills
 *       the others, then composites.  sp.P must equal svgir_params.P; sp.subset / sp.subset_count are ignored (the library's own list,
 *       kept in the geometry blob, is used).
 *   svgir_backward: after the rasterizer's backward has produced dL_dfeatures / dL_dvfeatures, differentiates the shading of the
 *       surfels with out_weights > 0 (all others have exactly-zero feature gradients) into svgir_grads.dL_dbase_color ...; the rest of
 *       those rows is zero-filled.  Pass the SAME struct (same `features` / `vfeatures` contents) as to the forward.
 * all_surfels != 0 shades / differentiates all P surfels (needed when `reduced` feeds a loss of its own, e.g. the reference's
 * lambda_light term, svgss.py:359-364).  Results are bit-identical to the unfused sequence (svgir_shade_forward -> svgir_forward,
 * svgir_backward -> svgir_shade_backward) except dL_denv, whose float atomics are summed in a different order. */
struct svgir_fused_shade {
    svgir_shade_params sp;
    float* reduced;        /* [P,70] or NULL; rows of unshaded surfels are zero */
    int32_t all_surfels;
};

/* Materialises the incident-direction lattice: dirs [P,Ns,3] and / or areas [P,Ns,1] (either may be NULL), the return
 * values of the reference's fibonacci_sphere_sampling(normals, Ns, random_rotate) with `offsets` [P] standing for its
 * random draw (NULL = random_rotate False).  lattice_work: scratch of 4 * Ns floats. */
int svgir_incident_dirs(int32_t P, int32_t Ns, const float* normals, const float* offsets, float* lattice_work,
                        float* dirs, float* areas, void* stream);

/* Bilinear resize of an [H,W,C] image to [out_h,out_w,C] with the half-pixel convention of
 * F.interpolate(mode='bilinear', align_corners=False): EnvLight.direct_light's 32x64 down-sample of its HDR map
 * (scene/envmap.py:62-63). */
int svgir_resample_bilinear(const float* src, int32_t H, int32_t W, int32_t C, float* dst, int32_t out_h, int32_t out_w,
                            void* stream);

/* ---- image-space epilogue right after the svgss rasterizer ----------------------------------
 * Fused replacement of the PyTorch tail of render_view (gaussian_renderer/svgss.py:187-246): feature / vfeature planes
 * divided by the rendered opacity (clamp_min 1e-5), channel split, rgb_to_srgb (utils/graphics_utils.py:198-215),
 * compositing over `bg` [3].  opacity [1,H,W], feature [S,H,W], vfeature [VS/4,H,W] are the rasterizer's outputs at the
 * training (S=4, VS=52) or evaluation (S=7, VS=64) widths; `out` receives svgir_unpack_planes(training) planes [.,H,W]:
 *   training  (21): pbr | normal | base_color | roughness | diffuse | local_lights | visibility
 *   evaluation(27): pbr | normal | base_color | roughness | direct | indirect | lights | local_lights | visibility
 * (three planes per quantity; one-channel quantities are broadcast over the three background channels as in the
 * reference).  The backward maps dL/d(out) to dL/d(opacity, feature, vfeature); all outputs are overwritten. */
int svgir_unpack_planes(int32_t training);
int svgir_unpack_forward(int32_t W, int32_t H, int32_t training, const float* bg, const float* opacity, const float* feature,
                         const float* vfeature, float* out, void* stream);
int svgir_unpack_backward(int32_t W, int32_t H, int32_t training, const float* bg, const float* opacity, const float* feature,
                          const float* vfeature, const float* dL_dout, float* dL_dopacity, float* dL_dfeature,
                          float* dL_dvfeature, void* stream);
/* Stage 1 (rgss) counterparts (gaussian_renderer/render.py:83-91, 107-114).
 * pack: features [P,5] = [geometric normal (world) 3, view depth d, d^2], d = (xyz1 @ viewmatrix).z; the backward maps
 *       dL/d(features) to dL/d(means3D) and dL/d(normals).
 * unpack: x_c = feature_c / max(opacity, 1e-5) * (num_contrib > 0) for the 5 feature planes; out [6,H,W] = [x_0 .. x_4,
 *       depth_var = x_4 - depth^2]; the backward maps dL/d(out) to dL/d(opacity, depth, feature). */
int svgir_pack_rgss_forward(int32_t P, const float* means3D, const float* normals, const float* viewmatrix, float* features,
                            void* stream);
int svgir_pack_rgss_backward(int32_t P, const float* means3D, const float* viewmatrix, const float* dL_dfeatures,
                             float* dL_dmeans3D, float* dL_dnormals, void* stream);
int svgir_unpack_rgss_forward(int32_t W, int32_t H, const int32_t* num_contrib, const float* opacity, const float* depth,
                              const float* feature, float* out, void* stream);
int svgir_unpack_rgss_backward(int32_t W, int32_t H, const int32_t* num_contrib, const float* opacity, const float* depth,
                               const float* feature, const float* dL_dout, float* dL_dopacity, float* dL_ddepth,
                               float* dL_dfeature, void* stream);

/* depth2normal (utils/image_utils.py:61-125): depth, mask [1,H,W] -> normal [3,H,W]; fovx / fovy in radians, prcp = the
 * camera's principal point as a fraction of the image (Camera.prcppoint). */
int svgir_depth2normal(int32_t W, int32_t H, const float* depth, const float* mask, float fovx, float fovy, float prcp_x,
                       float prcp_y, float* normal, void* stream);
/* adjoint of svgir_depth2normal w.r.t. the depth plane (the reference's depth2normal is differentiable and the stage-1 loss
 * `cos_loss(rendered_normal, d2n)` uses it without detaching, gaussian_renderer/render.py:158-160): dL_ddepth [1,H,W] is
 * written completely; the mask gets no gradient. */
int svgir_depth2normal_backward(int32_t W, int32_t H, const float* depth, const float* mask, const float* dL_dnormal, float fovx,
                                float fovy, float prcp_x, float prcp_y, float* dL_ddepth, void* stream);

/* Image losses behind render_view: L1 and SSIM with the reference's 11 x 11 Gaussian window
 * (`F.l1_loss(image, gt)` and `ssim(image, gt)`, gaussian_renderer/svgss.py:281-289, render.py:150-151;
 * utils/loss_utils.py:21-64), img1 / img2 = [C,H,W] planes.
 *   forward : writes 2 floats per 16x16 tile and channel -- the tile's sums of the SSIM map and of |img1 - img2| -- into
 *             `partial` [svgir_l1_ssim_partials(C,H,W)][2], if `dmaps` [3,C,H,W] is not NULL what the backward needs, and if
 *             `means2` is not NULL the two means {mean SSIM, mean |img1 - img2|} (device floats: fixed-order double sum of
 *             the partials by a second small kernel);
 *   backward: dL_dimg1 [C,H,W] = g_ssim_mean * d(mean SSIM)/d(img1) + g_l1_mean * d(mean |img1 - img2|)/d(img1), written
 *             completely; if `g_dev` is not NULL the two host scalars are multiplied by g_dev[0] / g_dev[1] read on the
 *             device (upstream gradients that never visit the host).  img2 (the ground truth) gets no gradient. */
size_t svgir_l1_ssim_partials(int32_t C, int32_t H, int32_t W);
int svgir_l1_ssim_forward(const float* img1, const float* img2, int32_t C, int32_t H, int32_t W, float* partial, float* dmaps,
                          float* means2, void* stream);
int svgir_l1_ssim_backward(const float* img1, const float* img2, const float* dmaps, int32_t C, int32_t H, int32_t W,
                           float g_ssim_mean, float g_l1_mean, const float* g_dev, float* dL_dimg1, void* stream);

/* The consumers of the rasterizer's gradients: Adam over the per-Gaussian parameter block, the
 * densification statistics, and the row compaction behind pruning (scene/gaussian_model.py:737-773, 1020-1062, 1270-1276).
 *
 * svgir_adam_step: one launch for up to SVGIR_ADAM_MAX_TENSORS parameter tensors -- torch.optim.Adam's update (amsgrad
 *   off, no weight decay) with a learning rate and a step count per tensor (`step` = the count AFTER this update, >= 1;
 *   bias corrections in double precision on the host, like torch).  Tensors with n = 0 are skipped.
 * svgir_densify_stats: GaussianModel.add_densification_stats -- weights_accum += weights (if given);
 *   for rows with update_filter: xyz_gradient_accum += |viewspace_grad[:, :2]|, denom += 1.  grad_stride = floats per row
 *   of viewspace_grad (3 in the reference).
 * svgir_mask_scan: order-preserving list `kept` [<= P] of the rows with keep != 0 and their number (count_dev[0], device
 *   memory); `work` = svgir_mask_scan_work_words(P) uint32 of scratch.
 * svgir_gather_rows: dst[t][r] = src[t][kept[r]] for r < min(rows_max, *count_dev), for up to SVGIR_ADAM_MAX_TENSORS
 *   tensors with rows of row_bytes (multiple of 4) in one launch: the `tensor[mask]` of _prune_optimizer for every
 *   parameter, both Adam moments and the bookkeeping arrays at once. */
#define SVGIR_ADAM_MAX_TENSORS 32
/* svgir_adam_tensor.flags -- GaussianModel.step() (scene/gaussian_model.py:775-813) folded into the Adam pass:
 *   SVGIR_ADAM_SCRUB_NAN : replace_nangrad_to_zero -- NaN gradient entries are replaced by `nan_value` (0 or 1e-6 by group)
 *                          before the update, and in the gradient tensor itself;
 *   SVGIR_ADAM_ZERO_GRAD : optimizer.zero_grad() -- the gradient tensor is left zero-filled. */
#define SVGIR_ADAM_SCRUB_NAN 1
#define SVGIR_ADAM_ZERO_GRAD 2
typedef struct svgir_adam_tensor {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;        /* elements */
    double lr;
    int32_t step;
    int32_t flags;    /* SVGIR_ADAM_* (the gradient is written when a flag is set) */
    float nan_value;
} svgir_adam_tensor;
typedef struct svgir_row_tensor { const void* src; void* dst; int32_t row_bytes; } svgir_row_tensor;
int svgir_adam_step(const svgir_adam_tensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream);
int svgir_densify_stats(int32_t P, const float* viewspace_grad, int32_t grad_stride, const uint8_t* update_filter,
                        const float* weights, float* weights_accum, float* xyz_gradient_accum, float* denom, void* stream);
size_t svgir_mask_scan_work_words(int32_t P);
int svgir_mask_scan(int32_t P, const uint8_t* keep, uint32_t* kept, uint32_t* work, uint32_t* count_dev, void* stream);
int svgir_gather_rows(const svgir_row_tensor* tensors, int32_t count, const uint32_t* kept, const uint32_t* count_dev,
                      int32_t rows_max, void* stream);

/* The producer of the visibility the shading kernels stream: a linear BVH over the surfels and the
 * visibility tracer of submodules/bvh (`RayTracer`, __init__.py:28-71; `create_bvh` src/bvh.cu:8-28 + src/construct.cu:151-265;
 * `trace_bvh_opacity` src/bvh.cu:86-115 + src/trace.cu:186-262).
 *   svgir_bvh_bytes : size of the opaque BVH blob for P surfels.
 *   svgir_bvh_build : leaf boxes from the eight corners mean +- 3 (s_a a +- s_b b +- s_c c) (a, b, c = columns of the
 *       rotation of the normalised quaternion), Morton sort, hierarchy, bottom-up refit.  means3D [P,3], scales [P,3],
 *       rotations [P,4] (r,x,y,z; need not be normalised).
 *   svgir_bvh_trace_visibility : per ray, origin rays_o + t_offset * rays_d (the reference's caller passes 0.05), the
 *       product of (1 - opacity exp(power)) over the surfels whose leaf box the ray enters, with opacity >= 1/255, facing
 *       the ray (normal . d <= 0), t >= 0.01 and power <= 0, where t = (m^T C d) / (d^T C d) and
 *       power = -1/2 (mean - x(t))^T C (mean - x(t)); visibility = 0 (and contribute = 0) as soon as the product drops
 *       below 0.9, else the product and the number of contributing surfels.  means3D [P,3], cov_inv [P,6] (symmetric,
 *       xx xy xz yy yz zz), opacity [P], normals [P,3] may differ from the arrays the BVH was built from (they do in the
 *       reference: build from scaling/rotation, trace with the inverse covariance); rays_o / rays_d [num_rays,3];
 *       contribute [num_rays] int32, visibility [num_rays]. */
size_t svgir_bvh_bytes(int32_t P);
int svgir_bvh_build(int32_t P, const float* means3D, const float* scales, const float* rotations, char* bvh, void* stream);
int svgir_bvh_trace_visibility(int32_t P, char* bvh, int64_t num_rays, const float* rays_o, const float* rays_d, float t_offset,
                               const float* means3D, const float* cov_inv, const float* opacity, const float* normals,
                               int32_t* contribute, float* visibility, void* stream);

/* The producer of the incident-radiance cache (the other half of row f3): the linear BVH and the closest-hit radiance
 * tracer of the reference's point-based-GI renderer (pbgi/bvhhelpers.py:96-156 `get_gs_bvh`; pbgi/renderer.py:582-615
 * `build_bvh`, `render_radiance_with_sampling_SH`; the slang kernels under pbgi/bvhworkers/: get_elements,
 * lbvh_morton_codes, lbvh_single_radixsort, lbvh_hierarchy, lbvh_bounding_boxes, intersect_test:1879-1990, sh_utils),
 * called by GaussianModel.update_radiace (scene/gaussian_model.py:469-522).
 *   svgir_pbgi_bvh_bytes  : size of the opaque blob for P >= 1 primitives.
 *   svgir_pbgi_bvh_build  : boxes centre +- 3 max|scale| (centers [P,3], scales [P,3]), 30-bit Morton codes of the box
 *       centres in the scene extent, stable sort by code, Karras hierarchy (equal codes resolved by sorted position),
 *       bottom-up box unions.  The tree is the reference's tree node for node (the tracer's result depends on it).
 *   svgir_pbgi_bvh_export : the tree as the reference's tensors: info [2P-1][3] int32 = {left, right, primitive}
 *       (`LBVHNode_info`), aabb [2P-1][6] (`LBVHNode_aabb`); optionally the sorted (code, primitive) pairs [P][2].
 *   svgir_pbgi_trace_radiance : N rows x S rays; ray_o [N,3] (one origin per row), ray_d [N,S,3].  Per ray: repeated
 *       closest-hit queries over t in [0.042 (0.01 after the first hit), 0.2] from the moving origin; each accepted hit adds
 *       eval_sh(degree 3, shs [P,16,3], direction origin -> hit centre) * alpha * T and multiplies T by (1 - alpha), until
 *       T <= 0.001, no hit, or the hit is the primitive whose index equals the ROW of the ray.  Outputs: radiance [N,S,3]
 *       clamped to [0,10], visibility [N,S] (T, or 0 once T < 0.2), hit_indices [N,S] int32 (first hit or -1), uvs [N,S,2].
 *       rotations [P,4] (r,x,y,z), normals [P,3], opacity [P], cov3D_inverse [P,6] (xx xy xz yy yz zz).
 *       oracle/pbgi_oracle.cpp lists the reference's traversal quirks that are reproduced.
 *       The blob is NOT read-only during a trace: the per-call leaf records, the row order of the call and its ray queue live in it.
 *       Calls that share one blob (build, trace) must therefore be ordered on ONE stream (or by events); concurrent traces from several
 *       streams / threads need one blob each (a build is 6 launches). */
size_t svgir_pbgi_bvh_bytes(int32_t P);
int svgir_pbgi_bvh_build(int32_t P, const float* centers, const float* scales, char* bvh, void* stream);
int svgir_pbgi_bvh_export(int32_t P, char* bvh, int32_t* info, float* aabb, int32_t* sorted, void* stream);
int svgir_pbgi_trace_radiance(int32_t P, char* bvh, int32_t N, int32_t S, const float* ray_o, const float* ray_d, const float* centers,
                              const float* scales, const float* rotations, const float* normals, const float* opacity,
                              const float* cov3D_inverse, const float* shs, float* radiance, float* visibility, int32_t* hit_indices,
                              float* uvs, void* stream);

/* Densification (scene/gaussian_model.py:1064-1248).
 * svgir_densify_masks : the selection of densify_and_clone / densify_and_split from the statistics densify_and_prune
 *   derives (grads = xyz_gradient_accum / denom and normal_gradient_accum / denom, NaN -> 0): selected when
 *   |grads| >= grad_threshold or |grads_normal| >= normal_threshold; clone_mask = selected and max(exp(scaling_raw)) <=
 *   size_limit (= percent_dense * scene_extent), split_mask = selected and > size_limit.  normal_gradient_accum may be NULL.
 * svgir_append_rows : cat_tensors_to_optimizer / densification_postfix for up to SVGIR_ADAM_MAX_TENSORS per-row tensors in
 *   one launch: dst = cat(src[0:rows_old], repeat(src[list[0:n]], repeat)), n = min(n_sel_max, *count_dev) (list / count_dev
 *   as produced by svgir_mask_scan); SVGIR_APPEND_ZERO_NEW leaves the new rows zero (the Adam moments of the new points).
 *   dst holds rows_old + n_sel_max * repeat rows of row_bytes (multiple of 4).
 * svgir_split_transform : densify_and_split's new points, in place on the n_new appended copies of the selected rows:
 *   xyz <- R(rotation) (z * exp(scaling)) + xyz, scaling <- log(exp(scaling) / (0.8 N)) with the last axis at -1e10;
 *   z [n_new,3] = the standard-normal draws (torch.normal(mean = 0, std = stds) = stds * z). */
#define SVGIR_APPEND_ZERO_NEW 1
typedef struct svgir_append_tensor { const void* src; void* dst; int32_t row_bytes; int32_t flags; } svgir_append_tensor;
int svgir_densify_masks(int32_t P, const float* xyz_gradient_accum, const float* normal_gradient_accum, const float* denom,
                        const float* scaling_raw, float grad_threshold, float normal_threshold, float size_limit,
                        uint8_t* clone_mask, uint8_t* split_mask, void* stream);
int svgir_append_rows(const svgir_append_tensor* tensors, int32_t count, int64_t rows_old, const uint32_t* list,
                      const uint32_t* count_dev, int64_t n_sel_max, int32_t repeat, void* stream);
int svgir_split_transform(int64_t n_new, int32_t N, const float* z, float* xyz_new, float* scaling_new, const float* rotation_new,
                          void* stream);

/* Per-stage GPU timing.  While enabled, forward/backward record HIP events on the launch stream at every stage
 * boundary (no extra synchronisation); svgir_last_timings() waits for the recorded events and returns, per stage,
 * the AVERAGE duration in milliseconds and the number of samples since profiling was (re-)enabled.
 * Stage names: "preprocess","sort_depth","scan","emit","sort_tile","ranges","render","image",
 *              "render_bwd","geom_bwd".  Returns the number of entries written (<= cap); `counts` may be NULL. */
void svgir_set_profiling(int enabled);
int svgir_last_timings(const char** names, float* avg_ms, int* counts, int cap);

const char* svgir_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SVGIR_RASTER_H_ */
