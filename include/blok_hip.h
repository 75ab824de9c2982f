/*
 * blok_hip.h — C ABI of the MI355X (gfx950) voxel ray-march backend for blok.
 *
 * This is the drop-in boundary.  blok has no plugin ABI; a backend is whatever
 * `App` calls on it (reference blok/src/app.cpp:73-128,130-192) and whatever
 * `Renderer::addWorld` consumes (reference blok/include/renderer.hpp:40-72).
 * Each entry point below names the reference interface it stands in for.
 *
 * Conventions
 *   - every function returns BLOK_OK (0) or a negative blok_status; the text of the
 *     last failure is available from blok_hip_last_error(ctx) (ctx may be NULL for
 *     create-time failures).  The reference throws std::runtime_error instead
 *     (reference blok/src/main.cpp:19-22); the C++ HipTracer wrapper rethrows.
 *   - plain pointers and sizes only; caller owns every host array, the context owns
 *     every device copy (same ownership split as reference renderer.hpp:195 +
 *     renderer_init.cpp:123-169).
 *   - a context is bound to one device and is not thread-safe (the reference is
 *     single-threaded: one context, one frame in flight on the compute path,
 *     reference blok/src/cuda_tracer.cu:539).
 *   - there is NO CPU fallback: without a gfx950 device / code object every entry
 *     point that would compute fails with BLOK_ERR_NO_DEVICE.
 */
#ifndef BLOK_HIP_H
#define BLOK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ records */

/* = blok::SvoNode, reference blok/include/svo.hpp:23-28.  16 B.
 * childMask bit i set <=> child i non-empty (reference blok/src/svo.cpp:99);
 * leaf <=> childMask == 0 (reference assets/shaders/intersect.rint:136). */
typedef struct blok_svo_node {
    uint32_t child_mask;
    uint32_t first_child; /* chunk-relative index of child 0, 8 siblings contiguous; 0xFFFFFFFF none */
    uint32_t material_id;
    float    occupancy;   /* >0 filled */
} blok_svo_node;

/* = blok::SubChunkGpu, reference blok/include/resources.hpp:170-184.  48 B. */
typedef struct blok_sub_chunk {
    uint32_t node_offset;     /* start of the parent chunk's nodes in the global array */
    uint32_t root_node_index; /* relative to node_offset */
    uint32_t node_count;      /* nodes in the parent chunk */
    uint32_t start_depth;
    float    world_min[3];
    float    sub_chunk_size;
    float    world_max[3];
    float    pad0;
} blok_sub_chunk;

/* = blok::MaterialGpu, reference blok/include/material.hpp:88-114.  32 B.
 * flags = metal<<24 | rough<<16 | type<<12 | alpha4<<8 | spec. */
typedef struct blok_material {
    float    albedo[3];
    uint32_t flags;
    float    emission[3];
    float    ior;
} blok_material;

/* = CameraCUDA, the reference's own compute-backend camera contract
 * (reference blok/src/cuda_tracer.cu:51-58, filled at :404-415 from blok::Camera,
 * reference blok/include/camera.hpp:25-42). */
typedef struct blok_camera {
    float pos[3];
    float fwd[3];
    float right[3];
    float up[3];
    float tan_half_fov; /* tanf(0.5*fov) */
    float aspect;       /* width / height of the FULL frame */
} blok_camera;

/* First-hit record, 16 B per ray.  What closest-hit sees of a procedural hit in the
 * reference: gl_HitTEXT, hitAttribs.materialId, gl_HitKindEXT
 * (reference assets/shaders/intersect.rint:138-141, hit.rchit:58-74), plus the voxel
 * that produced it.  Miss: t = -1 (reference miss.rmiss:25-27), hit = 0, face = 0xFF. */
typedef struct blok_hit {
    float    t;
    uint32_t material_id;
    int16_t  voxel[3];    /* world voxel coordinates of the leaf */
    uint8_t  face;        /* 0:+X 1:-X 2:+Y 3:-Y 4:+Z 5:-Z  (reference hit.rchit:46-53) */
    uint8_t  hit;         /* 1 hit, 0 miss */
} blok_hit;

/* Explicit ray, for secondary rays and edge-case tests (same interval semantics as
 * traceRayEXT: reference assets/shaders/raygen.rgen:217-229). */
typedef struct blok_ray {
    float org[3];
    float tmin;
    float dir[3];
    float tmax;
} blok_ray;

typedef enum blok_status {
    BLOK_OK              =  0,
    BLOK_ERR_INVALID_ARG = -1,
    BLOK_ERR_NO_DEVICE   = -2, /* no gfx950 device, or the HIP runtime failed */
    BLOK_ERR_HIP         = -3, /* a HIP call failed; message has the call and code */
    BLOK_ERR_NO_WORLD    = -4, /* trace before upload */
    BLOK_ERR_UNSUPPORTED = -5, /* world not on the unit-voxel integer lattice, leaf above voxel level, ... */
    BLOK_ERR_OOM         = -6
} blok_status;

/* Reference constants of the primary trace (reference assets/shaders/raygen.rgen:225,227). */
#define BLOK_RAY_TMIN 0.001f
#define BLOK_RAY_TMAX 10000.0f

typedef struct blok_hip_ctx blok_hip_ctx;

/* ---------------------------------------------------------------- lifecycle */

/* = CudaTracer::CudaTracer(w,h) + init()   (reference blok/include/cuda_tracer.hpp:25-28,
 *   blok/src/cuda_tracer.cu:425-448).  Binds the context to `device_ordinal`. */
int blok_hip_create(blok_hip_ctx** out_ctx, int device_ordinal, uint32_t width, uint32_t height);

/* = CudaTracer::resize (reference blok/src/cuda_tracer.cu:557-575). */
int blok_hip_resize(blok_hip_ctx* ctx, uint32_t width, uint32_t height);

/* = CudaTracer::shutdown / Renderer::cleanupWorld (reference blok/src/cuda_tracer.cu:577-602,
 *   blok/src/renderer_init.cpp:123-169). */
void blok_hip_destroy(blok_hip_ctx* ctx);

const char* blok_hip_last_error(const blok_hip_ctx* ctx);

/* A caller that is about to destroy a HIP stream it has passed to *_device entries tells the context first: the context keeps scratch
 * buffers and launch markers per stream (so that frames in flight on different streams never share any) and would otherwise keep them
 * until blok_hip_destroy.  Blocks until the device is idle.  No reference counterpart (the reference renders on one queue). */
int blok_hip_release_stream(blok_hip_ctx* ctx, void* hip_stream);

/* ------------------------------------------------------------------- world */

/* ChunkManager's voxelSize for the NEXT blok_hip_upload_world (reference blok/src/chunk_manager.cpp:19-25: voxel coordinates
 * times voxelSize are the world coordinates in every SubChunkGpu).  Default 1 (the reference app, app.cpp:37).  Powers of two
 * in [1/256, 256] are supported: the structure is built on the voxel lattice and the walk scales its integer planes, every
 * product exact, so first hits stay bit-exact; hit records carry the voxel's lattice coordinates.  Any other size returns
 * BLOK_ERR_UNSUPPORTED: its box planes are not exactly representable, and the reference's own result then depends on the
 * rounding of every intermediate box, which no other traversal order reproduces.  The shadow rays' last-occluder map and the
 * resident volume exist for size 1 only. */
int blok_hip_set_voxel_size(blok_hip_ctx* ctx, float voxel_size);

/* = Renderer::addWorld / updateWorld  (reference blok/include/renderer.hpp:40-72,
 *   uploadSvoBuffers + uploadMaterialBuffer, blok/src/renderer_upload.cpp:237-312).
 * Takes the three host arrays of WorldSvoGpu (reference blok/include/resources.hpp:195-203)
 * exactly as packChunksToGpuSvo emits them (reference blok/src/chunk_manager.cpp:234-314),
 * copies them to HBM and builds the traversal structure there (this also replaces
 * buildChunkBlas/buildChunkTlas, reference blok/src/renderer_raytracing.cpp:15-254).
 * Replaces any previous world. */
int blok_hip_upload_world(blok_hip_ctx* ctx,
                          const blok_svo_node* nodes, size_t n_nodes,
                          const blok_sub_chunk* sub_chunks, size_t n_sub_chunks,
                          const blok_material* materials, size_t n_materials);

/* Dense form (BASELINE.json configs[0..1]): material_ids[x + y*nx + z*nx*ny], 0 = empty,
 * i.e. the reference's dense store with density>0 <=> id != 0
 * (reference blok/include/chunk.hpp:35-36, blok/src/chunk_manager.cpp:57-59,330-348).
 * origin = world coordinate of voxel (0,0,0). */
int blok_hip_upload_dense(blok_hip_ctx* ctx, const uint32_t* material_ids,
                          uint32_t nx, uint32_t ny, uint32_t nz, const int32_t origin[3],
                          const blok_material* materials, size_t n_materials);
/* Dense-grid path (BASELINE.json configs[1]: "256^3 dense grid ... primary-ray DDA ... coalesced HBM, no SVO"; no reference
 * counterpart, the reference has no DDA): when on at the time of blok_hip_upload_dense, the grid itself stays on the device in
 * 8x8x8-cell tiles with one occupancy bit per tile (staged into LDS by the kernel), and blok_hip_trace_primary* (rectangle
 * entries) walk it with a two-level DDA over the same canonical plane sequence instead of the derived tree; the tile / ray /
 * path entries keep using the tree.  Records are identical either way (tests/test_gpu_parity.py).  Default off. */
int blok_hip_set_dense_dda(blok_hip_ctx* ctx, int enabled);

/* Sizes of the device-resident world, for accounting (bytes). */
typedef struct blok_world_stats {
    uint64_t n_voxels;          /* filled leaves */
    uint64_t n_ref_nodes;       /* reference SvoNode records uploaded */
    uint64_t n_sub_chunks;
    uint64_t n_tree_nodes;      /* 16-B 4x4x4 nodes of the derived structure */
    uint64_t tree_bytes;        /* nodes + material side array */
    uint32_t levels;            /* 4^levels voxels per axis */
    int32_t  origin[3];         /* world coordinate of the structure's corner */
} blok_world_stats;
int blok_hip_world_stats(const blok_hip_ctx* ctx, blok_world_stats* out);

/* -------------------------------------------------------------------- trace */

/* = CudaTracer::drawFrame(cam, ...) restricted to the primary hit
 *   (reference blok/src/cuda_tracer.cu:484-555) == raygen.rgen bounce 0 / sample 0
 *   -> traceRayEXT -> intersect.rint -> hit.rchit
 *   (reference assets/shaders/raygen.rgen:194-229).
 * Traces the pixel rectangle [x0,x0+w) x [y0,y0+h) of the ctx-sized frame (tile-able for
 * the multi-GPU partition) and copies the w*h records row-major to `out_hits_host`.
 * Blocking, like the reference's cudaDeviceSynchronize (cuda_tracer.cu:539). */
int blok_hip_trace_primary(blok_hip_ctx* ctx, const blok_camera* cam,
                           uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                           blok_hit* out_hits_host);

/* Same work, device-resident outputs, asynchronous on `hip_stream` (a hipStream_t, or NULL for the default
 * stream).  `out_hits_dev` (w*h 16-B records) and `out_rgba_dev` (w*h RGBA8 pixels: the frame through
 * hit.rchit's material fetch, reference assets/shaders/hit.rchit:58-67; the CUDA backend's output format,
 * reference blok/src/cuda_tracer.cu:385-386) are device pointers; either may be NULL, not both.
 * This is the entry the benchmark times and the one a graph capture may record (no allocation, no host
 * sync inside). */
int blok_hip_trace_primary_device(blok_hip_ctx* ctx, const blok_camera* cam,
                                  uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                  void* out_hits_dev, void* out_rgba_dev, void* hip_stream);

/* Interleaved-tile form of the multi-GPU partition: rank r of n traces the tiles
 * (tile x tile pixels, row-major tile index i) with i % n == r and writes them densely,
 * tile after tile, each tile row-major; each output holds
 * blok_hip_tiles_for_rank(...) * tile*tile elements (edge tiles are padded with misses / sky). */
uint32_t blok_hip_tiles_for_rank(uint32_t width, uint32_t height, uint32_t tile,
                                 uint32_t rank, uint32_t n_ranks);
int blok_hip_trace_tiles_device(blok_hip_ctx* ctx, const blok_camera* cam,
                                uint32_t tile, uint32_t rank, uint32_t n_ranks,
                                void* out_hits_dev, void* out_rgba_dev, void* hip_stream);
/* Root side: scatter n_ranks gathered tile buffers (rank-major, each padded to `tiles_per_rank_max`
 * tiles) of `elem_bytes`-sized elements (16: hit records, 4: RGBA8) back into a row-major
 * width*height frame. */
int blok_hip_untile_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t elem_bytes, uint32_t tile,
                           uint32_t n_ranks, uint32_t tiles_per_rank_max,
                           void* out_frame_dev, void* hip_stream);

/* Sparse framebuffer exchange for the tile partition (no reference counterpart: blok is single-GPU): most
 * of a frame's tiles can be sky, and only bytes cross xGMI slowly.  compact: from a rank's dense RGBA8 tile buffer (as
 * blok_hip_trace_tiles_device writes it) to  word 0 = number of tiles with at least one non-sky pixel, then one record
 * {local tile index, tile*tile pixels} per such tile, in no particular order; `out_words_dev` holds
 * blok_hip_compact_words(tile, n_tiles) 32-bit words.  A prefix of 1 + S * (1 + tile*tile) words carries the first S records.
 * scatter (on the root): writes the context's width x height RGBA8 frame: the tiles the ranks' records hold (rank r's buffer
 * starts at word r * rank_stride_words; at most max_records records each are looked at; local tile index j of rank r is frame
 * tile r + j * n_ranks) and the sky colour everywhere else.  Both asynchronous on `hip_stream`. */
size_t blok_hip_compact_words(uint32_t tile, uint32_t n_tiles);
int blok_hip_compact_tiles_device(blok_hip_ctx* ctx, const void* rgba_tiles_dev, uint32_t tile, uint32_t n_tiles,
                                  void* out_words_dev, void* hip_stream);
int blok_hip_scatter_tiles_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t n_ranks, size_t rank_stride_words,
                                  uint32_t tile, uint32_t max_records, void* out_frame_rgba_dev, void* hip_stream);

/* Several frames per call (no reference counterpart: the reference issues one traceRaysKHR per frame,
 * blok/src/renderer_raytracing.cpp:666-685, on one GPU).  At N ranks one frame's share of the tiles is 1/N of a launch whose
 * duration is mostly latency, and every call and collective costs host time, so a rank traces up to BLOK_MAX_TILE_FRAMES
 * consecutive frames — cams[0 .. n_frames), one camera each — in ONE beam + trace launch pair and exchanges them together.
 * Frame f of a dense tile buffer starts `frame_stride_tiles` tiles behind frame f - 1; the root's output frames are contiguous
 * (width*height elements each).  Each call equals n_frames calls of the one-frame
 * entry above it, bit for bit. */
#define BLOK_MAX_TILE_FRAMES 8
int blok_hip_trace_tile_frames_device(blok_hip_ctx* ctx, const blok_camera* cams, uint32_t n_frames,
                                      uint32_t tile, uint32_t rank, uint32_t n_ranks, uint32_t frame_stride_tiles,
                                      void* out_hits_dev, void* out_rgba_dev, void* hip_stream);
/* gathered: rank r's block starts at tile r * tiles_per_rank_max, frame f inside it at tile f * frame_stride_tiles */
int blok_hip_untile_frames_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t elem_bytes, uint32_t tile,
                                  uint32_t n_ranks, uint32_t tiles_per_rank_max, uint32_t n_frames, uint32_t frame_stride_tiles,
                                  void* out_frames_dev, void* hip_stream);
/* Sparse exchange of n_frames frames.  compact: `out_words_dev` holds n_frames * blok_hip_compact_words(tile, n_tiles) words:
 * n_frames count words, then the records INTERLEAVED by frame — record slot j of frame f at word
 * n_frames + (j * n_frames + f) * (1 + tile*tile) — so that the first S slots of all frames are one contiguous prefix of
 * n_frames * (1 + S * (1 + tile*tile)) words: what travels (n_frames = 1 is the one-frame layout above).
 * scatter (root): rank r's prefix starts at word r * rank_stride_words; writes n_frames contiguous width*height RGBA8 frames: a
 * tile some record holds gets its pixels, every other tile the sky colour, no pixel is written twice.  tile_state_dev (optional):
 * n_frames * ceil(width/tile) * ceil(height/tile) bytes that belong to `out_frames_rgba_dev` — zero while the buffer is all sky —
 * in which the call keeps "this tile of this frame buffer holds something other than sky"; with it a sky tile that stays sky is
 * not written at all (the root's assembly is on every frame's critical path).  NULL: every tile is written. */
int blok_hip_compact_tile_frames_device(blok_hip_ctx* ctx, const void* rgba_tiles_dev, uint32_t tile, uint32_t n_tiles,
                                        uint32_t n_frames, uint32_t frame_stride_tiles, void* out_words_dev, void* hip_stream);
int blok_hip_scatter_tile_frames_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t n_ranks, size_t rank_stride_words,
                                        uint32_t tile, uint32_t max_records, uint32_t n_frames,
                                        void* out_frames_rgba_dev, void* tile_state_dev, void* hip_stream);

/* The same exchange with 16 bits per pixel.  A shaded pixel is a function of (material id, face) or the sky, and every rank holds
 * the material table, so a pixel can travel as  code = min(material id, n_materials) * 8 + face  (0xFFFF = sky), made from the
 * rank's FIRST-HIT tiles (what blok_hip_trace_tile(_frame)s_device writes to out_hits_dev) and expanded on the root through the
 * function the trace kernels shade with: the assembled RGBA8 frames are bit-identical, the bytes on the wire half.
 * blok_hip_exchange_code_bits: 16 if the uploaded material table allows it ((n_materials + 1) * 8 <= 0xFFFF), else 0 (use the
 * RGBA8 entries).  Buffers as above with records of 1 + tile*tile/2 words: blok_hip_compact_code_words(tile, n_tiles) words per frame. */
uint32_t blok_hip_exchange_code_bits(const blok_hip_ctx* ctx);
size_t blok_hip_compact_code_words(uint32_t tile, uint32_t n_tiles);
int blok_hip_compact_hit_tile_frames_device(blok_hip_ctx* ctx, const void* hit_tiles_dev, uint32_t tile, uint32_t n_tiles,
                                            uint32_t n_frames, uint32_t frame_stride_tiles, void* out_words_dev, void* hip_stream);
int blok_hip_scatter_code_tile_frames_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t n_ranks, size_t rank_stride_words,
                                             uint32_t tile, uint32_t max_records, uint32_t n_frames,
                                             void* out_frames_rgba_dev, void* tile_state_dev, void* hip_stream);

/* The reference's per-pixel sample / bounce loop and G-buffer: raygen.rgen:167-414 with hit.rchit, miss.rmiss
 * and shadow.rmiss (reference assets/shaders/).  Planes are float4 per pixel of the rectangle, row-major; any
 * pointer may be NULL.  color = (rgb, 1); world_pos = (first-hit position, depth); normal_roughness;
 * albedo_metallic (raygen.rgen:392-407; the reference stores the last two as RGBA16F / RGBA8).
 * spp = FrameUBO.sampleCount (8 in the reference, renderer_denoising.cpp:683), max_bounces = MAX_BOUNCES
 * (2, raygen.rgen:211), frame_index = FrameUBO.frameCount (seeds the RNG, raygen.rgen:92-99). */
typedef struct blok_gbuffer {
    float* color;
    float* world_pos;
    float* normal_roughness;
    float* albedo_metallic;
} blok_gbuffer;
int blok_hip_trace_paths_device(blok_hip_ctx* ctx, const blok_camera* cam,
                                uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                                const blok_gbuffer* planes_dev, void* hip_stream);
/* The same frame with the G-buffer in the reference's own image formats (raygen.rgen:55-59; created at
 * blok/src/renderer_denoising.cpp:110-170): colour RGBA32F, world position + depth RGBA32F, normal + roughness RGBA16F, albedo +
 * metallic RGBA8 (unorm), motion vectors RG16F written by the path kernel itself (computeMotionVector, raygen.rgen:150-155,
 * 409-413, from prev_view_proj = FrameUBO::prevViewProj, column-major; 0 on the sky).  48 B/pixel instead of 64.  Any plane may
 * be NULL; a motion plane needs prev_view_proj.  Conversions as the image stores do them: binary16 round to nearest even, unorm8
 * = floor(clamp(x, 0, 1) * 255 + 0.5).  blok_hip_denoise_ref_device takes these planes; blok_hip_draw_frame_rt uses them. */
typedef struct blok_gbuffer_ref {
    float*    color;             /* RGBA32F */
    float*    world_pos;         /* RGBA32F: xyz, depth */
    uint16_t* normal_roughness;  /* RGBA16F */
    uint32_t* albedo_metallic;   /* RGBA8: r | g << 8 | b << 16 | metallic << 24 */
    uint16_t* motion;            /* RG16F */
} blok_gbuffer_ref;
int blok_hip_trace_paths_ref_device(blok_hip_ctx* ctx, const blok_camera* cam,
                                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                    uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                                    const float prev_view_proj[16], const blok_gbuffer_ref* planes_dev, void* hip_stream);
/* Blocking form with host planes. */
int blok_hip_trace_paths(blok_hip_ctx* ctx, const blok_camera* cam,
                         uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                         uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                         const blok_gbuffer* planes_host);

/* = the tonemap pass of the post-process chain (reference assets/shaders/tonemap.comp:97-143, dispatched at
 * blok/src/renderer_postprocess.cpp:588-612): HDR float4 -> RGBA8.  Reference defaults: exposure 1.0,
 * saturation_boost 1.15, tonemap_operator 1 = Khronos PBR Neutral (0 = soft clip)
 * (reference blok/include/renderer_postprocess.hpp:110-113). */
int blok_hip_tonemap_device(blok_hip_ctx* ctx, const float* hdr_rgba_dev, uint32_t n_pixels, float exposure,
                            float saturation_boost, int tonemap_operator, void* out_rgba8_dev, void* hip_stream);
int blok_hip_tonemap(blok_hip_ctx* ctx, const float* hdr_rgba_host, uint32_t n_pixels, float exposure,
                     float saturation_boost, int tonemap_operator, uint32_t* out_rgba8_host);

/* Explicit rays (secondary rays; edge-case tests). n rays in, n records out, host arrays. */
int blok_hip_trace_rays(blok_hip_ctx* ctx, const blok_ray* rays_host, size_t n,
                        blok_hit* out_hits_host);

/* Shade first hits to RGBA8 (normal/albedo debug view through hit.rchit's material fetch,
 * reference assets/shaders/hit.rchit:55-76): out_rgba8_host[w*h]. */
int blok_hip_shade_rgba8(blok_hip_ctx* ctx, const blok_camera* cam,
                         uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                         uint32_t* out_rgba8_host);

/* = CudaTracer::drawFrame in the compute backend's progressive mode (reference blok/src/cuda_tracer.cu:484-555 with the
 * kernel's accumulate / ACES / gamma tail, :372-386, :209-216, :95-99): traces spp_per_frame samples per pixel of the
 * reference's sample/bounce loop with the RNG frame index = frames accumulated so far, adds the frame's average to the
 * accumulation buffer (xyz running sum, w = frames), clears that buffer first when any camera component moved by more
 * than 1e-5 (camChanged, :456-472) and writes the ACES-tonemapped, gamma-2.2 RGBA8 image of the running average to
 * out_rgba8_host (may be NULL).  Blocking. */
int blok_hip_draw_frame_accumulate(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t spp_per_frame, uint32_t max_bounces,
                                   uint32_t* out_rgba8_host, uint32_t* out_frames_accumulated);
/* The accumulation buffer: width*height float4 (xyz sum, w frames). */
int blok_hip_accum_download(blok_hip_ctx* ctx, float* out_rgba32f_host);
/* = CudaTracer::resetAccum (reference blok/src/cuda_tracer.cu:450-454). */
int blok_hip_reset_accum(blok_hip_ctx* ctx);

/* Milliseconds of the most recent trace kernel, measured with HIP events recorded on the
 * launch stream around the kernel only (valid after that stream has been synchronised). */
int blok_hip_last_kernel_ms(blok_hip_ctx* ctx, float* out_ms);

/* ---- image-space chain behind the path tracer -----------------------------------------------
 * Denoiser (temporal accumulation -> variance -> a-trous iterations), TAA resolve and sharpen as HIP kernels; the tonemap
 * pass between TAA and sharpen is blok_hip_tonemap_device.  Reference: assets/shaders/temporal_reproject.comp, variance.comp,
 * atrous.comp, taa.comp, sharpen.comp; orchestration blok/src/renderer_denoising.cpp:714-776, renderer_postprocess.cpp:505-556.
 * All planes are full-frame (ctx width x height), row-major, in device memory; calls enqueue on hip_stream and return. */
typedef struct blok_denoise_settings {       /* = Denoiser::Settings, blok/include/renderer_denoising.hpp:49-66 */
    float temporal_alpha, moment_alpha, variance_clip_gamma;
    float depth_threshold, normal_threshold;
    float phi_color, phi_normal, phi_depth;
    int32_t atrous_iterations;               /* <= 5 (DenoiserPipeline::MAX_ATROUS_ITERATIONS) */
    float variance_boost;
    int32_t min_history_length;
} blok_denoise_settings;
void blok_denoise_settings_default(blok_denoise_settings* out);
/* = Denoiser::denoise + copyCurrentGeometryToHistory + swapHistoryBuffers (renderer_denoising.cpp:714-776, 833-866, 690-697)
 * for one frame.  planes: float4 planes as blok_hip_trace_paths_device writes them (color, world_pos, normal_roughness;
 * albedo_metallic unused).  motion_dev: float2 per pixel, or NULL = the motion vectors of raygen.rgen:150-155,409-413 computed
 * from world_pos and prev_view_proj (column-major 4x4, = FrameUBO::prevViewProj).  frame_count = FrameUBO::frameCount
 * (0: no history is read).  settings NULL = defaults.  out_color_dev: float4 per pixel = Denoiser::getOutputImage(). */
int blok_hip_denoise_device(blok_hip_ctx* ctx, const blok_gbuffer* planes_dev, const float* motion_dev,
                            const float prev_view_proj[16], uint32_t frame_count, const blok_denoise_settings* settings,
                            float* out_color_dev, void* hip_stream);
/* The same for planes in the reference's image formats (blok_gbuffer_ref; motion NULL = computed from world_pos). */
int blok_hip_denoise_ref_device(blok_hip_ctx* ctx, const blok_gbuffer_ref* planes_dev, const float prev_view_proj[16],
                                uint32_t frame_count, const blok_denoise_settings* settings, float* out_color_dev, void* hip_stream);
/* Host copies of the denoiser's state after the last blok_hip_denoise_device (blocking; NULL pointers are skipped):
 * history colour float4, moments float2, history length float, variance float, motion vectors float2 (per pixel). */
int blok_hip_denoise_state(blok_hip_ctx* ctx, float* history_color, float* moments, float* history_length,
                           float* variance, float* motion);
/* = PostProcess TAA pass (taa.comp; renderer_postprocess.cpp:526-534,558-589): color_dev float4 in, out_color_dev float4 out;
 * the TAA history is kept by the context.  motion_dev: float2 per pixel or NULL = the motion vectors of the last
 * blok_hip_denoise_device call.  Defaults of the reference: feedback_min 0.93, feedback_max 0.98. */
int blok_hip_taa_device(blok_hip_ctx* ctx, const float* color_dev, const float* motion_dev, float feedback_min,
                        float feedback_max, uint32_t frame_count, float* out_color_dev, void* hip_stream);
/* = PostProcess sharpen pass (sharpen.comp; renderer_postprocess.cpp:548-555,619-642): RGBA8 in, RGBA8 out; default strength 0.5. */
int blok_hip_sharpen_device(blok_hip_ctx* ctx, const uint32_t* rgba8_dev, float strength, uint32_t* out_rgba8_dev, void* hip_stream);
/* Column-major (GLM layout) view-projection of a camera basis, such that ndc.xy * 0.5 + 0.5 of (M * vec4(p, 1)) is the screen
 * position of world point p under the basis' own primary-ray mapping: what FrameUBO::prevViewProj is to the reference's
 * shaders (temporal_reproject.comp:108-113), for callers that keep camera bases instead of matrices. */
void blok_camera_view_proj(const blok_camera* cam, float out_view_proj[16]);
/* = Renderer::drawFrame's ray-tracing path (reference blok/src/renderer_draw.cpp: trace -> Denoiser::denoise -> PostProcess::process):
 * one frame of the sample/bounce loop (spp samples, the reference forces 8; max_bounces, the reference's MAX_BOUNCES is 2) into
 * planes owned by the context, then denoiser, TAA (feedback 0.93..0.98), tonemap (Khronos PBR neutral, exposure 1, saturation
 * 1.15) and sharpen (0.5) with the reference's default settings; the previous frame's camera supplies prevViewProj and the frame
 * counter (RNG frame index, FrameUBO::frameCount) advances by one per call; blok_hip_post_reset restarts it.  Blocking.
 * out_rgba8_host: width*height RGBA8, may be NULL. */
int blok_hip_draw_frame_rt(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t spp, uint32_t max_bounces,
                           const blok_denoise_settings* settings, uint32_t* out_rgba8_host, uint32_t* out_frame_count);
/* Forget the denoiser and TAA histories (swapchain recreate: Denoiser::resize / PostProcess::resize). */
int blok_hip_post_reset(blok_hip_ctx* ctx);

/* ---- device-resident dense voxel store (edits and the rebuild they trigger, on the GPU) ----
 * The reference keeps Chunk::density / Chunk::materialIds on the host (blok/src/chunk.hpp:33-42), edits them with
 * setVoxelMaterial (chunk_manager.cpp:316-328) and applyBrush (brush.cpp:13-63), rebuilds dirty chunks' SVOs on the CPU
 * (chunk_manager.cpp:106-140), repacks (:213-314) and re-uploads (renderer_upload.cpp:237-312).  Here one box of the world
 * [origin, origin + (nx, ny, nz)) lives in HBM with the same two arrays (x fastest, then y, then z), edits are kernels
 * with the reference's float arithmetic, and blok_hip_volume_rebuild derives the traversal structure on the device from
 * "density > 0" (chunk_manager.cpp:121) and installs it as the context's world.  voxel_size must be 1; chunk_size is
 * ChunkManager's chunk edge (the brush computes voxel centres per chunk).  Edits outside the box are refused
 * (BLOK_ERR_UNSUPPORTED, nothing written).  All calls are blocking. */
int blok_hip_volume_create(blok_hip_ctx* ctx, const int32_t origin[3], uint32_t nx, uint32_t ny, uint32_t nz,
                           uint32_t chunk_size, float voxel_size);
int blok_hip_volume_destroy(blok_hip_ctx* ctx);
/* Whole-box upload / download of the two arrays; a null pointer uploads zeros / skips the download. */
int blok_hip_volume_upload(blok_hip_ctx* ctx, const float* density, const uint32_t* material_ids);
int blok_hip_volume_download(blok_hip_ctx* ctx, float* density, uint32_t* material_ids);
/* = ChunkManager::setVoxelMaterial for n world voxels xyz[3*i..], in order (a later entry of the same voxel wins);
 * material_ids null = 0, density null = 1. */
int blok_hip_volume_set_voxels(blok_hip_ctx* ctx, const int32_t* xyz, const uint32_t* material_ids, const float* density, size_t n);
/* = applyBrush (brush.cpp:13-63): mode 0 ADD density = max(density, value), 1 SUBTRACT density = min(density, value),
 * inside the sphere around `center` (world units), voxel centres as the reference computes them. */
int blok_hip_volume_apply_brush(blok_hip_ctx* ctx, const float center[3], float radius, float value, int mode);
/* = rebuildDirtyChunks + packChunksToGpuSvo + Renderer::updateWorld for the box: installs the world made of the voxels with
 * density > 0 and their material ids, with the given material table. */
int blok_hip_volume_rebuild(blok_hip_ctx* ctx, const blok_material* materials, size_t n_materials);

/* TAA jitter of the primary rays of all following frames, in pixels (each within +-0.5; NULL or {0,0} = none, the default and
 * the parity / benchmark contract).  The reference applies its Halton(2,3) - 0.5 sequence through the projection matrix
 * (getJitteredProjection, blok/src/renderer_postprocess.cpp:254-268: proj[2][0..1] += 2 j / size, handed to raygen.rgen as
 * invProj, blok/src/renderer_draw.cpp:64-81); in the basis form of this backend that is the same ray as NDC + 2 j / size, i.e. a
 * sub-pixel offset of +j pixels.  blok_taa_jitter (blok_world.h) gives the sequence.  blok_hip_draw_frame_rt applies entry
 * (frame mod 16) by itself (blok_hip_set_rt_taa_jitter(ctx, 0) = PostProcess::Settings::enableTAA false for the jitter). */
int blok_hip_set_taa_jitter(blok_hip_ctx* ctx, const float jitter_px[2]);
int blok_hip_set_rt_taa_jitter(blok_hip_ctx* ctx, int enabled);
/* Enable/disable the per-launch HIP event pair (default off: nothing but the kernel is
 * enqueued by the *_device entries). */
int blok_hip_set_timing(blok_hip_ctx* ctx, int enabled);

/* Library/ABI version: (major<<16)|minor. */
uint32_t blok_hip_abi_version(void);

/* ------------------------------------------------------------- several devices, one process
 * The tile partition of the frame over the GPUs of one node driven from one host thread (no reference
 * counterpart — blok is single-GPU): one context and one stream per device in `device_ordinals` (the first is the root), the
 * world replicated on each, every device traces tiles rank, rank + G, ... of the tile x tile grid, the RGBA8 tiles are gathered
 * on the root and un-permuted into the row-major frame; first-hit records stay on the device that traced them.
 * Transport: RCCL when allow_rccl != 0, librccl is found at run time (dlopen; no link-time dependency) and the devices are
 * distinct — one communicator per device (ncclCommInitAll), one ncclGroupStart/End per frame with every peer's ncclSend and the
 * root's ncclRecvs, each on its device's stream; otherwise hipMemcpyPeerAsync from every peer into the root's buffer (also how
 * a one-GPU box rehearses several ranks: the same ordinal may be listed more than once); one device: none.
 * blok_hip_multi_transport() says which ("rccl", "peer-copy", "none").  The per-device contexts are ordinary contexts
 * (blok_hip_multi_context) for settings such as blok_hip_set_beam. */
typedef struct blok_hip_multi blok_hip_multi;
int  blok_hip_multi_create(blok_hip_multi** out, const int* device_ordinals, uint32_t n_devices, uint32_t width, uint32_t height,
                           uint32_t tile, int allow_rccl);
void blok_hip_multi_destroy(blok_hip_multi* m);
const char* blok_hip_multi_last_error(const blok_hip_multi* m);      /* NULL: the last failed create on this thread */
uint32_t blok_hip_multi_device_count(const blok_hip_multi* m);
const char* blok_hip_multi_transport(const blok_hip_multi* m);
blok_hip_ctx* blok_hip_multi_context(blok_hip_multi* m, uint32_t rank);
/* = Renderer::addWorld on every device (replicated). */
int  blok_hip_multi_upload_world(blok_hip_multi* m, const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* sub_chunks,
                                 size_t n_sub_chunks, const blok_material* materials, size_t n_materials);
/* One frame, asynchronous: enqueues trace, exchange and un-permute; *out_rgba8_dev_on_root (may be NULL) = the root's
 * width x height RGBA8 frame, valid after blok_hip_multi_synchronize.  blok_hip_multi_draw_frame = that + synchronise + copy. */
int  blok_hip_multi_draw_frame_device(blok_hip_multi* m, const blok_camera* cam, const uint32_t** out_rgba8_dev_on_root);
int  blok_hip_multi_synchronize(blok_hip_multi* m);
int  blok_hip_multi_draw_frame(blok_hip_multi* m, const blok_camera* cam, uint32_t* out_rgba8_host);
/* Rank `rank`'s first-hit records of the last synchronised frame, in its tile order (blok_hip_tiles_for_rank x tile^2). */
/* Several frames per call (1..BLOK_MAX_TILE_FRAMES cameras, one launch pair per device; the frames lie one after the other in
 * the root's buffer / in out_rgba8_host), and how the tiles reach the root:
 *   "sparse-pull" (mode 1; the default, mode -1, whenever it can be had): every device compacts its tiles with a hit into 16-bit
 *       (material, face) code records in its own memory and the root's assembly kernel reads counts and records straight out of
 *       the peers' memory through peer mappings and expands them — nothing staged, no size on the host, only live records cross
 *       a link, bit-identical frames.  Needs peer access from the root to every device (refused with BLOK_ERR_UNSUPPORTED
 *       otherwise) and a material table that fits the codes (else the call uses "dense").
 *   "dense" (mode 0): the RGBA8 tiles of every rank travel whole (blok_hip_multi_transport), then an un-permute kernel. */
int  blok_hip_multi_set_exchange(blok_hip_multi* m, int mode);
const char* blok_hip_multi_exchange(const blok_hip_multi* m);        /* what the next call will use: "sparse-pull" or "dense" */
int  blok_hip_multi_draw_frames_device(blok_hip_multi* m, const blok_camera* cams, uint32_t n_frames, const uint32_t** out_rgba8_dev_on_root);
int  blok_hip_multi_draw_frames(blok_hip_multi* m, const blok_camera* cams, uint32_t n_frames, uint32_t* out_rgba8_host);
/* first-hit records of the first frame of the last call */
int  blok_hip_multi_download_hits(blok_hip_multi* m, uint32_t rank, blok_hit* out_host, size_t capacity_records);

#ifdef __cplusplus
}
#endif
#endif /* BLOK_HIP_H */
