// Material library and MagicaVoxel .vox import behind include/blok_world.h.
//
// Follows the reference's behaviour (blok/src/material.cpp, blok/include/material.hpp:88-114,
// blok/src/vox_loader.cpp:116-462) with its own structure: the file is parsed from a memory cursor whose
// "failed" state mirrors an std::ifstream's (after one short read every later read fails too), the default
// palette is generated from its construction rule instead of a 256-entry table, and truncated XYZI chunks stop
// at the last complete voxel (the reference keeps pushing uninitialised voxels there).
#include "blok_world.h"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

enum : uint8_t { kDiffuse = 0, kMetallic = 1, kGlass = 2, kEmissive = 3 };

inline float clamp01(float v) { return std::min(std::max(v, 0.0f), 1.0f); }

void set_error(char* err, size_t n, const std::string& msg) {
    if (err && n) { std::snprintf(err, n, "%s", msg.c_str()); }
}

// MagicaVoxel's default palette: entry 0 empty; 215 colours of the 6x6x6 cube over {ff,cc,99,66,33,00}
// (blue fastest, then green, then red, black left out); then ten-step ramps of red, green, blue and grey.
// Values are ABGR.  (The reference carries the same values as a literal table, vox_loader.cpp:22-55;
// tests/test_vox.py compares against it when the reference tree is present.)
void default_palette(uint32_t pal[256]) {
    static const uint32_t cube[6] = {0xff, 0xcc, 0x99, 0x66, 0x33, 0x00};
    static const uint32_t ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    uint32_t k = 0;
    pal[k++] = 0;
    for (uint32_t r : cube) for (uint32_t g : cube) for (uint32_t b : cube)
        if (r | g | b) pal[k++] = 0xff000000u | (b << 16) | (g << 8) | r;
    for (uint32_t v : ramp) pal[k++] = 0xff000000u | v;
    for (uint32_t v : ramp) pal[k++] = 0xff000000u | (v << 8);
    for (uint32_t v : ramp) pal[k++] = 0xff000000u | (v << 16);
    for (uint32_t v : ramp) pal[k++] = 0xff000000u | (v << 16) | (v << 8) | v;
}

struct Cursor {
    const uint8_t* data;
    size_t size, pos = 0;
    bool ok = true;
    bool read(void* dst, size_t n) {
        if (!ok) return false;
        if (n > size - pos) { pos = size; ok = false; return false; }     // short read poisons the stream
        std::memcpy(dst, data + pos, n);
        pos += n;
        return true;
    }
    template <class T> bool value(T& v) { return read(&v, sizeof(T)); }
    void seek(size_t p) { if (ok) pos = std::min(p, size); }
    void skip(int64_t n) { if (ok) pos = static_cast<size_t>(std::min<int64_t>(std::max<int64_t>(int64_t(pos) + n, 0), int64_t(size))); }
};

struct VoxMaterial {                       // vox_loader.hpp:26-37
    uint8_t type = kDiffuse;
    float roughness = 0.5f, metallic = 0.0f, ior = 1.5f, emission = 0.0f, flux = 0.0f, alpha = 1.0f, glow = 0.0f,
          specular = 0.5f;
    bool has_properties = false;
};
struct VoxModel { uint32_t size[3] = {0, 0, 0}; std::vector<uint8_t> voxels; };   // 4 bytes per voxel

std::string read_string(Cursor& c) {       // vox_loader.cpp:68-74
    int32_t len = 0;
    if (!c.value(len) || len <= 0 || len > 1024) return "";
    std::string s(static_cast<size_t>(len), '\0');
    c.read(s.data(), s.size());
    return s;
}
float parse_float(const std::string& s, float fallback) {   // std::stof or the default, vox_loader.cpp:107-113
    errno = 0;
    char* end = nullptr;
    const float v = std::strtof(s.c_str(), &end);
    if (end == s.c_str() || errno == ERANGE) return fallback;
    return v;
}

}  // namespace

struct blok_material_library {
    std::vector<blok_material_desc> materials;
    std::unordered_map<std::string, uint32_t> by_name;
    std::unordered_map<uint32_t, uint32_t> by_color;
    uint32_t vox_palette[256];
    blok_material_library() { reset(); }
    void reset() {                                             // material.cpp:12-28,137-143
        materials.clear(); by_name.clear(); by_color.clear();
        std::fill(std::begin(vox_palette), std::end(vox_palette), 0u);
        blok_material_desc d;
        blok_material_desc_init(&d);
        std::snprintf(d.name, sizeof(d.name), "default");
        d.albedo[0] = d.albedo[1] = d.albedo[2] = 0.8f;
        add(d);
    }
    uint32_t add(const blok_material_desc& m) {                // material.cpp:30-39
        const uint32_t id = static_cast<uint32_t>(materials.size());
        materials.push_back(m);
        if (m.name[0]) by_name[m.name] = id;
        return id;
    }
};

struct blok_vox {
    std::vector<VoxModel> models;
    uint32_t palette[256];
    VoxMaterial materials[256];
};

namespace {

bool parse_vox(Cursor& c, blok_vox& out, std::string& err) {
    char magic[4];
    if (!c.read(magic, 4) || std::memcmp(magic, "VOX ", 4) != 0) { err = "Invalid VOX file: bad magic number"; return false; }
    int32_t version = 0;
    if (!c.value(version)) { err = "Failed to read VOX version"; return false; }
    if (version < 150) { err = "Unsupported VOX version: " + std::to_string(version) + " (need >= 150)"; return false; }
    default_palette(out.palette);
    out.models.clear();
    VoxModel current;
    bool has_size = false;
    char id[4];
    int32_t content = 0, children = 0;
    if (!c.read(id, 4) || std::memcmp(id, "MAIN", 4) != 0) { err = "Invalid VOX file: missing MAIN chunk"; return false; }
    if (!c.value(content) || !c.value(children)) { err = "Failed to read MAIN chunk header"; return false; }
    if (content > 0) c.skip(content);
    const int64_t end = int64_t(c.pos) + children;
    while (int64_t(c.pos) < end && c.ok) {
        if (!c.read(id, 4) || !c.value(content) || !c.value(children)) break;
        const size_t chunk_end = static_cast<size_t>(std::max<int64_t>(int64_t(c.pos) + content, 0));
        if (!std::memcmp(id, "SIZE", 4)) {
            if (has_size && !current.voxels.empty()) { out.models.push_back(std::move(current)); current = VoxModel{}; }
            int32_t x = 0, y = 0, z = 0;
            c.value(x); c.value(y); c.value(z);
            current.size[0] = uint32_t(x); current.size[1] = uint32_t(y); current.size[2] = uint32_t(z);
            has_size = true;
        } else if (!std::memcmp(id, "XYZI", 4)) {
            int32_t n = 0;
            if (!c.value(n)) { err = "Failed to read voxel count"; return false; }
            for (int32_t i = 0; i < n; ++i) {
                uint8_t v[4];
                if (!c.read(v, 4)) break;          // truncated: keep the complete voxels only
                current.voxels.insert(current.voxels.end(), v, v + 4);
            }
        } else if (!std::memcmp(id, "RGBA", 4)) {
            for (int i = 0; i < 255; ++i) {        // file entry i is palette index i+1; the last one is unused
                uint32_t rgba = 0;
                if (c.value(rgba)) out.palette[i + 1] = rgba;
            }
        } else if (!std::memcmp(id, "MATL", 4)) {
            int32_t material_id = 0;
            if (c.value(material_id)) {
                int32_t pairs = 0;
                std::unordered_map<std::string, std::string> props;
                if (c.value(pairs))
                    for (int32_t i = 0; i < pairs; ++i) {
                        std::string key = read_string(c);
                        std::string val = read_string(c);
                        if (!key.empty()) props[key] = val;
                    }
                if (material_id >= 0 && material_id < 256) {
                    VoxMaterial& m = out.materials[material_id];
                    m.has_properties = true;
                    auto get = [&](const char* k) -> const std::string* { auto it = props.find(k); return it == props.end() ? nullptr : &it->second; };
                    if (auto* s = get("_type"))
                        m.type = *s == "_metal" ? kMetallic : *s == "_glass" ? kGlass : *s == "_emit" ? kEmissive : kDiffuse;
                    if (auto* s = get("_rough")) m.roughness = parse_float(*s, 0.5f);
                    if (auto* s = get("_metal")) m.metallic = parse_float(*s, 0.0f);
                    if (auto* s = get("_ior")) m.ior = parse_float(*s, 1.5f);
                    if (auto* s = get("_emit")) m.emission = parse_float(*s, 0.0f);
                    if (auto* s = get("_flux")) m.flux = parse_float(*s, 0.0f);
                    if (auto* s = get("_alpha")) m.alpha = parse_float(*s, 1.0f);
                    if (auto* s = get("_sp")) m.specular = parse_float(*s, 0.5f);
                    if (auto* s = get("_g")) m.glow = parse_float(*s, 0.0f);
                }
            }
        }
        // like the reference's seekg pair; after a short read the stream stays failed and the loop ends
        c.seek(chunk_end);
        if (children > 0) c.skip(children);
    }
    if (has_size || !current.voxels.empty()) out.models.push_back(std::move(current));
    if (out.models.empty()) { err = "No models found in VOX file"; return false; }
    return true;
}

int load_from(Cursor c, blok_vox** out, char* err, size_t err_len) {
    auto* v = new (std::nothrow) blok_vox();
    if (!v) return BLOK_ERR_OOM;
    std::string why;
    if (!parse_vox(c, *v, why)) { set_error(err, err_len, why); delete v; return BLOK_ERR_INVALID_ARG; }
    *out = v;
    return BLOK_OK;
}

}  // namespace

extern "C" {

void blok_material_desc_init(blok_material_desc* m) {          // material.hpp:27-44
    if (!m) return;
    std::memset(m, 0, sizeof(*m));
    m->albedo[0] = m->albedo[1] = m->albedo[2] = 1.0f;
    m->alpha = 1.0f; m->metallic = 0.0f; m->roughness = 0.5f; m->ior = 1.5f; m->specular = 0.5f;
    m->emission_power = 0.0f; m->type = kDiffuse; m->vox_palette_index = -1;
}

void blok_material_pack(const blok_material_desc* m, blok_material* out) {   // material.hpp:96-112
    if (!m || !out) return;
    for (int a = 0; a < 3; ++a) out->albedo[a] = m->albedo[a];
    const uint32_t metal = static_cast<uint32_t>(clamp01(m->metallic) * 255.0f);
    const uint32_t rough = static_cast<uint32_t>(clamp01(m->roughness) * 255.0f);
    const uint32_t type = m->type;
    const uint32_t alpha = static_cast<uint32_t>(clamp01(m->alpha) * 15.0f);
    const uint32_t spec = static_cast<uint32_t>(clamp01(m->specular) * 255.0f);
    out->flags = (metal << 24) | (rough << 16) | (type << 12) | (alpha << 8) | spec;
    for (int a = 0; a < 3; ++a) out->emission[a] = m->emission[a] * m->emission_power;
    out->ior = m->type == kGlass ? m->ior : m->emission_power;
}

int blok_material_library_create(blok_material_library** out) {
    if (!out) return BLOK_ERR_INVALID_ARG;
    *out = new (std::nothrow) blok_material_library();
    return *out ? BLOK_OK : BLOK_ERR_OOM;
}
void blok_material_library_destroy(blok_material_library* lib) { delete lib; }
uint32_t blok_material_library_size(const blok_material_library* lib) { return lib ? uint32_t(lib->materials.size()) : 0; }
uint32_t blok_material_library_add(blok_material_library* lib, const blok_material_desc* m) { return lib && m ? lib->add(*m) : 0; }
uint32_t blok_material_library_add_or_find(blok_material_library* lib, const blok_material_desc* m) {   // material.cpp:41-49
    if (!lib || !m) return 0;
    if (m->name[0]) { auto it = lib->by_name.find(m->name); if (it != lib->by_name.end()) return it->second; }
    return lib->add(*m);
}
int blok_material_library_get(const blok_material_library* lib, uint32_t id, blok_material_desc* out) {   // material.cpp:51-63
    if (!lib || !out) return BLOK_ERR_INVALID_ARG;
    *out = lib->materials[id < lib->materials.size() ? id : 0];
    return BLOK_OK;
}
uint32_t blok_material_library_id_by_name(const blok_material_library* lib, const char* name) {   // material.cpp:73-79
    if (!lib || !name) return 0;
    auto it = lib->by_name.find(name);
    return it == lib->by_name.end() ? 0 : it->second;
}
uint32_t blok_material_library_from_color(blok_material_library* lib, uint8_t r, uint8_t g, uint8_t b) {   // material.cpp:81-117
    if (!lib) return 0;
    const uint32_t packed = (uint32_t(r) << 16) | (uint32_t(g) << 8) | uint32_t(b);
    auto it = lib->by_color.find(packed);
    if (it != lib->by_color.end()) return it->second;
    blok_material_desc m;
    blok_material_desc_init(&m);
    m.albedo[0] = float(r) / 255.0f; m.albedo[1] = float(g) / 255.0f; m.albedo[2] = float(b) / 255.0f;
    std::snprintf(m.name, sizeof(m.name), "color_%06X", packed);
    const uint32_t id = lib->add(m);
    lib->by_color[packed] = id;
    return id;
}
void blok_material_library_set_vox_palette(blok_material_library* lib, uint8_t i, uint32_t id) { if (lib) lib->vox_palette[i] = id; }
uint32_t blok_material_library_from_vox_palette(const blok_material_library* lib, uint8_t i) { return lib ? lib->vox_palette[i] : 0; }
int blok_material_library_pack(const blok_material_library* lib, blok_material* out, size_t capacity) {   // material.cpp:127-135
    if (!lib || !out || capacity < lib->materials.size()) return BLOK_ERR_INVALID_ARG;
    for (size_t i = 0; i < lib->materials.size(); ++i) blok_material_pack(&lib->materials[i], &out[i]);
    return BLOK_OK;
}
void blok_material_library_clear(blok_material_library* lib) { if (lib) lib->reset(); }

int blok_vox_load_memory(const void* data, size_t size, blok_vox** out, char* err, size_t err_len) {
    if (!out || (!data && size)) return BLOK_ERR_INVALID_ARG;
    *out = nullptr;
    return load_from(Cursor{static_cast<const uint8_t*>(data), size}, out, err, err_len);
}
int blok_vox_load_file(const char* path, blok_vox** out, char* err, size_t err_len) {
    if (!out || !path) return BLOK_ERR_INVALID_ARG;
    *out = nullptr;
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) { set_error(err, err_len, std::string("Failed to open file ") + path); return BLOK_ERR_INVALID_ARG; }
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return load_from(Cursor{bytes.data(), bytes.size()}, out, err, err_len);
}
void blok_vox_free(blok_vox* v) { delete v; }
uint32_t blok_vox_model_count(const blok_vox* v) { return v ? uint32_t(v->models.size()) : 0; }
int blok_vox_model_info(const blok_vox* v, uint32_t model, uint32_t size_xyz[3], uint32_t* n_voxels) {
    if (!v || model >= v->models.size()) return BLOK_ERR_INVALID_ARG;
    if (size_xyz) for (int a = 0; a < 3; ++a) size_xyz[a] = v->models[model].size[a];
    if (n_voxels) *n_voxels = uint32_t(v->models[model].voxels.size() / 4);
    return BLOK_OK;
}
const uint8_t* blok_vox_model_voxels(const blok_vox* v, uint32_t model) {
    return v && model < v->models.size() ? v->models[model].voxels.data() : nullptr;
}
const uint32_t* blok_vox_palette(const blok_vox* v) { return v ? v->palette : nullptr; }

int blok_vox_get_material(const blok_vox* v, uint8_t index, blok_material_desc* out) {   // vox_loader.cpp:116-149
    if (!v || !out) return BLOK_ERR_INVALID_ARG;
    blok_material_desc_init(out);
    const uint32_t c = v->palette[index];
    out->albedo[0] = float(c & 0xFF) / 255.0f;
    out->albedo[1] = float((c >> 8) & 0xFF) / 255.0f;
    out->albedo[2] = float((c >> 16) & 0xFF) / 255.0f;
    out->alpha = float((c >> 24) & 0xFF) / 255.0f;
    const VoxMaterial& m = v->materials[index];
    if (m.has_properties) {
        out->type = m.type; out->roughness = m.roughness; out->metallic = m.metallic; out->ior = m.ior;
        out->specular = m.specular; out->alpha = m.alpha;
        if (m.type == kEmissive) {
            for (int a = 0; a < 3; ++a) out->emission[a] = out->albedo[a];
            out->emission_power = m.emission > 0 ? m.emission : m.flux;
            if (out->emission_power <= 0) out->emission_power = 5.0f;
        }
    } else {
        out->type = kDiffuse; out->roughness = 0.5f; out->metallic = 0.0f;
    }
    out->vox_palette_index = int16_t(index);
    return BLOK_OK;
}

int blok_vox_import_materials(const blok_vox* v, blok_material_library* lib, uint32_t map[256]) {   // vox_loader.cpp:370-388
    if (!v || !lib) return BLOK_ERR_INVALID_ARG;
    for (int i = 1; i < 256; ++i) {
        blok_material_desc m;
        blok_vox_get_material(v, uint8_t(i), &m);
        std::snprintf(m.name, sizeof(m.name), "vox_mat_%d", i);
        const uint32_t id = lib->add(m);
        if (map) map[i] = id;
        lib->vox_palette[i] = id;
    }
    if (map) map[0] = 0;
    return BLOK_OK;
}

uint32_t blok_vox_import_to_world(const blok_vox* v, blok_world* w, const float offset[3], uint32_t model_index) {   // vox_loader.cpp:390-430
    if (!v || !w || model_index >= v->models.size()) return 0;
    const float o[3] = {offset ? offset[0] : 0.0f, offset ? offset[1] : 0.0f, offset ? offset[2] : 0.0f};
    const VoxModel& model = v->models[model_index];
    blok_material_library* lib = blok_world_get_material_library(w);
    uint32_t count = 0;
    for (size_t i = 0; i + 3 < model.voxels.size(); i += 4) {
        const uint8_t x = model.voxels[i], y = model.voxels[i + 1], z = model.voxels[i + 2], ci = model.voxels[i + 3];
        const float pos[3] = {o[0] + float(x), o[1] + float(z), o[2] + float(y)};    // VOX z (up) -> world y
        if (lib) blok_world_set_voxel(w, pos, lib->vox_palette[ci], 1.0f);
        else {
            const uint32_t c = v->palette[ci];
            blok_world_set_voxel_rgb(w, pos, uint8_t(c & 0xFF), uint8_t((c >> 8) & 0xFF), uint8_t((c >> 16) & 0xFF), 1.0f);
        }
        ++count;
    }
    return count;
}

int blok_load_and_import_vox(const char* path, blok_world* w, blok_material_library* lib, const float offset[3],
                             uint32_t model_index, char* err, size_t err_len) {   // vox_loader.cpp:432-462
    if (!w) return BLOK_ERR_INVALID_ARG;
    blok_vox* v = nullptr;
    const int rc = blok_vox_load_file(path, &v, err, err_len);
    if (rc != BLOK_OK) return rc;
    if (lib) {
        uint32_t map[256];
        blok_vox_import_materials(v, lib, map);
        blok_world_set_material_library(w, lib);
    }
    const uint32_t count = blok_vox_import_to_world(v, w, offset, model_index);
    blok_vox_free(v);
    if (count == 0) { set_error(err, err_len, "no voxels imported"); return BLOK_ERR_INVALID_ARG; }
    return BLOK_OK;
}

}  // extern "C"
