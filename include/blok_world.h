/*
 * blok_world.h — C ABI of the host-side voxel data model that feeds the trace path:
 * Morton codec, per-chunk sparse-voxel-octree build, sub-chunk packing, plus the
 * synthetic scene / camera helpers the benchmark uses.  Host code only (no GPU).
 *
 * Each entry names the reference interface it stands in for.  Byte layout of the
 * emitted records is the reference's (see blok_hip.h).  One documented deviation:
 * chunks are packed in sorted (cz,cy,cx) order; the reference iterates an
 * std::unordered_map (reference blok/src/chunk_manager.cpp:249), so its chunk order is
 * unspecified.  Node order inside a chunk is the reference's.
 */
#ifndef BLOK_WORLD_H
#define BLOK_WORLD_H

#include "blok_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------- morton */
/* = blok::morton3d::encode / decode / octantFromCode
 *   (reference blok/include/morton.hpp:23-33,46-53,55-58). */
uint64_t blok_morton_encode(int32_t x, int32_t y, int32_t z);
void     blok_morton_decode(uint64_t code, int32_t* x, int32_t* y, int32_t* z);
uint32_t blok_morton_octant(uint64_t code, uint32_t max_depth, uint32_t level);

/* ------------------------------------------------------------------ world */
typedef struct blok_world blok_world;

/* = ChunkManager::ChunkManager(C, voxelSize) (reference blok/src/chunk_manager.cpp:19-25).
 * chunk_size must be a power of two >= 8 (the reference app uses 128, app.cpp:37). */
int  blok_world_create(blok_world** out, uint32_t chunk_size, float voxel_size);
void blok_world_destroy(blok_world* w);
const char* blok_world_last_error(const blok_world* w);

/* = ChunkManager::setVoxelMaterial(worldPos, materialId, density)
 *   (reference blok/src/chunk_manager.cpp:316-328): floor -> chunk -> local, last write wins,
 *   marks the chunk dirty. */
int blok_world_set_voxel(blok_world* w, const float world_pos[3], uint32_t material_id, float density);
/* Bulk integer form of the same write (global voxel coordinates, density 1). */
int blok_world_set_voxels(blok_world* w, const int32_t* xyz, const uint32_t* material_ids, size_t n);
/* = ChunkManager::getVoxelMaterial (reference blok/src/chunk_manager.cpp:330-348). */
uint32_t blok_world_get_voxel_material(const blok_world* w, const float world_pos[3]);

/* = applyBrush(mgr, Brush{centerWS, radiusWS, value, mode}) (reference blok/src/brush.cpp:13-63,
 * blok/include/brush.hpp:14-21): sphere edit of the density field — mode 0 ADD: d = max(d, value),
 * mode 1 SUBTRACT: d = min(d, value); material ids are left as they are; every chunk the brush's bounding box
 * touches is created and every chunk with a voxel inside the sphere is marked dirty. */
int blok_world_apply_brush(blok_world* w, const float center[3], float radius, float value, int mode);

/* = rebuildDirtyChunks(mgr, maxPerFrame) (reference blok/src/chunk_manager.cpp:121-140):
 *   clear + re-insert every density>0 voxel in z,y,x order (x fastest, :106-119) through
 *   SvoTree::insertVoxel (reference blok/src/svo.cpp:59-101).  Returns chunks rebuilt (>=0). */
int blok_world_rebuild_dirty(blok_world* w, int max_per_frame);

/* = packChunksToGpuSvo (reference blok/src/chunk_manager.cpp:234-314). */
int blok_world_pack(blok_world* w);
size_t blok_world_node_count(const blok_world* w);
size_t blok_world_sub_chunk_count(const blok_world* w);
const blok_svo_node*  blok_world_nodes(const blok_world* w);
const blok_sub_chunk* blok_world_sub_chunks(const blok_world* w);

/* Per-chunk view (sorted order), for parity tests against the oracle's node arrays. */
size_t blok_world_chunk_count(const blok_world* w);
int    blok_world_chunk_info(const blok_world* w, size_t i, int32_t coord[3], uint64_t* n_nodes);
const blok_svo_node* blok_world_chunk_nodes(const blok_world* w, size_t i);
/* = SvoTree::findLeaf (reference blok/src/svo.cpp:103-130): node index in chunk i, or -1. */
int64_t blok_world_find_leaf(const blok_world* w, size_t i, uint32_t x, uint32_t y, uint32_t z);

/* -------------------------------------------------------------- materials */
/* = blok::Material (reference blok/include/material.hpp:27-44), C layout. */
typedef struct blok_material_desc {
    float   albedo[3];        /* default 1,1,1 */
    float   alpha;            /* 1 */
    float   metallic;         /* 0 */
    float   roughness;        /* 0.5 */
    float   ior;              /* 1.5 */
    float   specular;         /* 0.5 */
    float   emission[3];      /* 0 */
    float   emission_power;   /* 0 */
    uint8_t type;             /* MaterialType: 0 diffuse, 1 metallic, 2 glass, 3 emissive (material.hpp:18-24) */
    int16_t vox_palette_index;/* -1 */
    char    name[32];
} blok_material_desc;
void blok_material_desc_init(blok_material_desc* m);           /* the defaults above */
/* = MaterialGpu::pack (reference blok/include/material.hpp:96-112). */
void blok_material_pack(const blok_material_desc* m, blok_material* out);

/* = blok::MaterialLibrary (reference blok/include/material.hpp:116-163, blok/src/material.cpp):
 * id 0 is the default grey diffuse material; ids are indices into the packed GPU table. */
typedef struct blok_material_library blok_material_library;
int      blok_material_library_create(blok_material_library** out);
void     blok_material_library_destroy(blok_material_library* lib);
uint32_t blok_material_library_size(const blok_material_library* lib);
uint32_t blok_material_library_add(blok_material_library* lib, const blok_material_desc* m);          /* addMaterial */
uint32_t blok_material_library_add_or_find(blok_material_library* lib, const blok_material_desc* m);  /* addOrFindMaterial */
int      blok_material_library_get(const blok_material_library* lib, uint32_t id, blok_material_desc* out); /* getMaterial (id clamps to 0) */
uint32_t blok_material_library_id_by_name(const blok_material_library* lib, const char* name);        /* getMaterialIdByName */
uint32_t blok_material_library_from_color(blok_material_library* lib, uint8_t r, uint8_t g, uint8_t b); /* getOrCreateFromColor */
void     blok_material_library_set_vox_palette(blok_material_library* lib, uint8_t palette_index, uint32_t material_id);
uint32_t blok_material_library_from_vox_palette(const blok_material_library* lib, uint8_t palette_index);
int      blok_material_library_pack(const blok_material_library* lib, blok_material* out, size_t capacity); /* packForGpu */
void     blok_material_library_clear(blok_material_library* lib);
/* = ChunkManager::setMaterialLibrary (reference blok/include/chunk_manager.hpp:30): with a library
 * attached, colour writes go through getOrCreateFromColor, otherwise the id is r<<16|g<<8|b
 * (reference blok/src/chunk_manager.cpp:91-102). */
void blok_world_set_material_library(blok_world* w, blok_material_library* lib);
blok_material_library* blok_world_get_material_library(const blok_world* w);
int  blok_world_set_voxel_rgb(blok_world* w, const float world_pos[3], uint8_t r, uint8_t g, uint8_t b, float density);

/* ------------------------------------------------------------ .vox import */
/* MagicaVoxel .vox reader = loadVoxFile (reference blok/src/vox_loader.cpp:151-368): SIZE / XYZI / RGBA / MATL. */
typedef struct blok_vox blok_vox;
int  blok_vox_load_file(const char* path, blok_vox** out, char* err, size_t err_len);
int  blok_vox_load_memory(const void* data, size_t size, blok_vox** out, char* err, size_t err_len);
void blok_vox_free(blok_vox* v);
uint32_t blok_vox_model_count(const blok_vox* v);
int  blok_vox_model_info(const blok_vox* v, uint32_t model, uint32_t size_xyz[3], uint32_t* n_voxels);
/* voxels of a model as (x, y, z, colorIndex) bytes, file order */
const uint8_t* blok_vox_model_voxels(const blok_vox* v, uint32_t model);
const uint32_t* blok_vox_palette(const blok_vox* v);            /* 256 entries, ABGR */
/* = VoxFile::getMaterial (reference blok/src/vox_loader.cpp:116-149) */
int  blok_vox_get_material(const blok_vox* v, uint8_t palette_index, blok_material_desc* out);
/* = importVoxMaterials (reference blok/src/vox_loader.cpp:370-388) */
int  blok_vox_import_materials(const blok_vox* v, blok_material_library* lib, uint32_t palette_to_material[256]);
/* = importVoxToChunks (reference blok/src/vox_loader.cpp:390-430): VOX z is up -> world y. Returns voxels imported. */
uint32_t blok_vox_import_to_world(const blok_vox* v, blok_world* w, const float world_offset[3], uint32_t model_index);
/* = loadAndImportVox (reference blok/src/vox_loader.cpp:432-462); lib may be NULL. */
int  blok_load_and_import_vox(const char* path, blok_world* w, blok_material_library* lib,
                              const float world_offset[3], uint32_t model_index, char* err, size_t err_len);

/* ----------------------------------------------------------------- camera */
/* = blok::Camera::forward/right/up + toDevice(Camera,w,h)
 *   (reference blok/include/camera.hpp:25-42, blok/src/cuda_tracer.cu:404-415). */
int blok_camera_from_yaw_pitch(const float pos[3], float yaw_deg, float pitch_deg, float fov_deg,
                               uint32_t width, uint32_t height, blok_camera* out);
/* Look-at convenience: derives yaw/pitch, then as above. */
int blok_camera_look_at(const float pos[3], const float target[3], float fov_deg,
                        uint32_t width, uint32_t height, blok_camera* out);

/* The matrices the reference's Vulkan path hands its shaders (FrameUBO, blok/include/resources.hpp:103-150), from the same
 * basis record: view = glm::lookAt(pos, pos + forward, up) (blok/include/camera.hpp:49-52), proj = glm::perspective(fov,
 * aspect, near, far) with depth 0..1 and p[1][1] *= -1 (camera.hpp:9,54-59), inverse = glm::inverse (FrameUBO::invView /
 * invProj, blok/src/renderer_denoising.cpp:669-670).  glm itself is absent from the reference tree (empty submodule): these
 * follow its published formulas; all 4x4 are column-major floats (M[col * 4 + row]).
 * With them a pixel's ray is  normalize(invView * (normalize((invProj * (ndc, 1, 1)).xyz), 0))  (raygen.rgen:201-205) — the
 * same ray as the basis form the kernels use, and with blok_jittered_projection the same ray as the basis form with the
 * jitter as a sub-pixel offset (tests/test_host_model.py). */
void blok_camera_view(const blok_camera* cam, float out_view[16]);
void blok_camera_projection(const blok_camera* cam, float z_near, float z_far, float out_proj[16]);
int  blok_mat4_inverse(const float m[16], float out[16]);
/* TAA jitter of frame `frame_index` in pixels: entry frame_index mod 16 of the Halton(2,3) - 0.5 sequence
 * (PostProcess::initJitterSequence / halton / advanceJitter, blok/src/renderer_postprocess.cpp:208-228,243,660-663);
 * frame 0 = (0, -1/6).  blok_taa_jitter_clip = getJitterClipSpace (:234-241); blok_jittered_projection = getJitteredProjection
 * (:254-268): proj[2][0] += 2 jx / width, proj[2][1] += 2 jy / height. */
void blok_taa_jitter(uint32_t frame_index, float out_px[2]);
void blok_taa_jitter_clip(const float jitter_px[2], uint32_t width, uint32_t height, float out_clip[2]);
void blok_jittered_projection(const float proj[16], const float jitter_px[2], uint32_t width, uint32_t height, float out[16]);

/* ------------------------------------------------- synthetic benchmark scene */
/* Integer-only generator G(N, seed) of SURVEY.md §8(d): terrain shell + 64 shell spheres,
 * materialId in [1,255].  Writes into `w` (chunk size as created), then the caller
 * rebuilds and packs.  Not reference behaviour — benchmark input synthesis. */
int blok_scene_generate(blok_world* w, uint32_t n, uint32_t seed, uint64_t* out_n_voxels);
/* Same voxel set as a dense id grid ids[x + y*n + z*n*n] (0 = empty); n <= 1024 (4 GiB of ids). */
int blok_scene_generate_dense(uint32_t n, uint32_t seed, uint32_t* ids, uint64_t* out_n_voxels);
/* 256 hashed diffuse materials (roughness 0.5, metallic 0: reference material.cpp:102-106). */
int blok_scene_materials(uint32_t seed, blok_material* out256);
/* Poses A(0) outside-corner, B(1) inside-grazing, C(2) top-down of SURVEY.md §8(d). */
int blok_scene_camera(uint32_t n, uint32_t seed, int pose, uint32_t width, uint32_t height, blok_camera* out);

#ifdef __cplusplus
}
#endif
#endif /* BLOK_WORLD_H */
