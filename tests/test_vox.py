"""MagicaVoxel .vox import + material library, product vs oracle restatement
(reference blok/src/vox_loader.cpp, blok/src/material.cpp).  Files are synthesised here; the reference's own
.vox assets are only read (never copied) when the reference tree is present."""
import re
import struct
from pathlib import Path

import numpy as np
import pytest

from blok_amd import vox as V
from blok_amd import world as W
from blok_amd._ffi import BlokError
from tests import oracle_ffi as O

REF = Path("/root/reference")


def chunk(cid: bytes, content: bytes = b"", children: bytes = b"") -> bytes:
    return cid + struct.pack("<ii", len(content), len(children)) + content + children


def vox_string(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<i", len(b)) + b


def matl(material_id: int, props: dict) -> bytes:
    body = struct.pack("<ii", material_id, len(props))
    for k, v in props.items():
        body += vox_string(k) + vox_string(v)
    return chunk(b"MATL", body)


def model_chunks(size, voxels) -> bytes:
    xyzi = struct.pack("<i", len(voxels)) + b"".join(struct.pack("<4B", *v) for v in voxels)
    return chunk(b"SIZE", struct.pack("<iii", *size)) + chunk(b"XYZI", xyzi)


def make_vox(models, palette=None, materials=(), extra=b"", version=150) -> bytes:
    children = b"".join(model_chunks(s, v) for s, v in models)
    children += extra
    if palette is not None:
        children += chunk(b"RGBA", np.asarray(palette, dtype="<u4").tobytes())
    children += b"".join(matl(i, p) for i, p in materials)
    return b"VOX " + struct.pack("<i", version) + chunk(b"MAIN", b"", children)


def random_model(rng, size, n):
    xyz = rng.integers(0, size, size=(n, 3))
    ci = rng.integers(1, 256, size=(n, 1))
    return [tuple(int(v) for v in row) for row in np.concatenate([xyz, ci], axis=1)]


@pytest.fixture(scope="module")
def sample_file():
    rng = np.random.default_rng(42)
    pal = rng.integers(0, 1 << 32, size=256, dtype=np.uint64).astype(np.uint32)
    mats = [(3, {"_type": "_metal", "_rough": "0.15", "_metal": "0.9", "_sp": "0.8"}),
            (7, {"_type": "_emit", "_emit": "3.5", "_flux": "2"}),
            (9, {"_type": "_emit", "_flux": "1.25"}),
            (11, {"_type": "_emit"}),                                   # power defaults to 5 (vox_loader.cpp:139)
            (12, {"_type": "_glass", "_ior": "1.33", "_alpha": "0.25", "_rough": "not-a-number"}),
            (13, {"_type": "_weird", "_rough": "7.5", "_g": "0.5"}),    # unknown type -> diffuse; roughness clamps when packed
            (300, {"_type": "_metal"})]                                  # id outside 0..255: ignored
    unknown = chunk(b"nTRN", b"\x01\x02\x03\x04\x05", chunk(b"JUNK", b"zzzz"))     # skipped with its children
    data = make_vox([((40, 30, 20), random_model(rng, 20, 700)), ((8, 8, 8), random_model(rng, 8, 60))],
                    palette=pal, materials=mats, extra=unknown)
    return data


def test_parse_matches_oracle(sample_file):
    pv, ov = V.VoxFile.load_memory(sample_file), O.OracleVox(sample_file)
    assert ov.h is not None and pv.model_count() == ov.n_models() == 2
    for i in range(2):
        (ps, pvox), (os_, ovox) = pv.model(i), ov.model(i)
        assert ps == os_ and np.array_equal(pvox, ovox)
    assert np.array_equal(pv.palette(), ov.palette())
    assert pv.model(0)[0] == (40, 30, 20) and len(pv.model(0)[1]) == 700


def test_materials_and_gpu_table_match_oracle(sample_file):
    pv, ov = V.VoxFile.load_memory(sample_file), O.OracleVox(sample_file)
    plib, olib = V.MaterialLibrary(), O.OracleMaterialLibrary()
    pmap, omap = pv.import_materials(plib), olib.import_vox(ov)
    assert np.array_equal(pmap, omap) and len(plib) == len(olib) == 256 and pmap[0] == 0 and pmap[255] == 255
    pg, og = plib.pack_for_gpu(), olib.pack()
    assert pg.tobytes() == og.tobytes()
    # spot checks against the reference's rules
    assert np.allclose(pg[0]["albedo"], 0.8) and pg[0]["flags"] == (0 << 24) | (127 << 16) | (15 << 8) | 127   # default material
    assert (pg[3]["flags"] >> 12) & 15 == 1 and (pg[3]["flags"] >> 24) == int(0.9 * 255) and (pg[3]["flags"] >> 16) & 255 == int(0.15 * 255)
    assert np.allclose(pg[7]["emission"], pg[7]["albedo"] * 3.5) and pg[7]["ior"] == np.float32(3.5)            # ior slot = emissionPower
    assert np.allclose(pg[9]["emission"], pg[9]["albedo"] * 1.25)
    assert np.allclose(pg[11]["emission"], pg[11]["albedo"] * 5.0)
    assert (pg[12]["flags"] >> 12) & 15 == 2 and pg[12]["ior"] == np.float32(1.33) and (pg[12]["flags"] >> 16) & 255 == 127
    assert (pg[13]["flags"] >> 12) & 15 == 0 and (pg[13]["flags"] >> 16) & 255 == 255
    d = plib.get_material(7)[0]
    assert d["name"] == b"vox_mat_7" and d["vox_palette_index"] == 7 and d["type"] == 3
    assert plib.get_material_id_by_name("vox_mat_200") == 200 and plib.get_material_id_by_name("nope") == 0
    assert plib.get_material(100000)[0]["name"] == b"default"


@pytest.mark.parametrize("with_library", [True, False])
def test_import_to_chunks_matches_oracle(sample_file, with_library):
    """importVoxToChunks: VOX z is up (-> world y), offset, material ids from the palette map, or packed RGB
    without a library; the resulting SVO arrays must be byte-identical."""
    pv, ov = V.VoxFile.load_memory(sample_file), O.OracleVox(sample_file)
    cm, ow = W.ChunkManager(128, 1.0), O.OracleWorld(128, 1.0)
    plib = olib = None
    if with_library:
        plib, olib = V.MaterialLibrary(), O.OracleMaterialLibrary()
        pv.import_materials(plib); olib.import_vox(ov)
        cm._lib.blok_world_set_material_library(cm._h, plib._h)
    offset = (-20.0, 100.5, 117.0)                     # crosses chunk borders and the origin
    n_p = pv.import_to_chunks(cm, offset, 0)
    n_o = O.vox_import_to_world(ov, ow, olib, offset, 0)
    assert n_p == n_o == 700
    cm.rebuild_dirty_chunks(); ow.rebuild()
    pw = cm.pack_chunks_to_gpu_svo()
    on, osub = ow.pack()
    assert pw.nodes.tobytes() == on.tobytes() and pw.sub_chunks.tobytes() == osub.tobytes()
    x, y, z, ci = (int(v) for v in pv.model(0)[1][-1])
    want = ci if with_library else (lambda c: ((c & 255) << 16) | (((c >> 8) & 255) << 8) | ((c >> 16) & 255))(int(pv.palette()[ci]))
    assert cm.get_voxel_material((offset[0] + x, offset[1] + z, offset[2] + y)) == want
    assert pv.import_to_chunks(cm, offset, 5) == 0     # bad model index


def test_color_materials_match_oracle():
    plib, olib = V.MaterialLibrary(), O.OracleMaterialLibrary()
    for rgb in [(255, 0, 0), (1, 2, 3), (255, 0, 0), (0, 0, 0), (1, 2, 3), (200, 100, 50)]:
        assert plib.get_or_create_from_color(*rgb) == olib.from_color(*rgb)
    assert len(plib) == len(olib) == 5
    assert plib.pack_for_gpu().tobytes() == olib.pack().tobytes()
    assert plib.get_material(1)[0]["name"] == b"color_FF0000"
    plib.clear()
    assert len(plib) == 1


def test_default_palette_and_bad_files():
    rng = np.random.default_rng(1)
    data = make_vox([((4, 4, 4), random_model(rng, 4, 10))])           # no RGBA chunk -> default palette
    pv, ov = V.VoxFile.load_memory(data), O.OracleVox(data)
    assert np.array_equal(pv.palette(), ov.palette())
    pal = pv.palette()
    assert pal[0] == 0 and pal[1] == 0xFFFFFFFF and pal[2] == 0xFFCCFFFF and pal[215] == 0xFF330000 and pal[255] == 0xFF111111
    for bad, msg in [(b"", "magic"), (b"VOXX" + b"\0" * 20, "magic"), (b"VOX " + struct.pack("<i", 100), "version"),
                     (b"VOX " + struct.pack("<i", 150) + b"NOPE" + b"\0" * 8, "MAIN"),
                     (b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", b""), "No models")]:
        with pytest.raises(BlokError, match=msg):
            V.VoxFile.load_memory(bad)
        o = O.OracleVox(bad)
        assert o.h is None and msg in o.error
    with pytest.raises(BlokError, match="open"):
        V.VoxFile.load_file("/nonexistent/file.vox")
    truncated = data[:-7]                                                # cut inside the XYZI payload
    assert len(V.VoxFile.load_memory(truncated).model(0)[1]) == 8        # complete voxels only


def test_load_and_import_file(tmp_path, sample_file):
    path = tmp_path / "model.vox"
    path.write_bytes(sample_file)
    cm, lib = W.ChunkManager(128, 1.0), V.MaterialLibrary()
    V.load_and_import_vox(path, cm, lib)
    assert len(lib) == 256 and cm.rebuild_dirty_chunks() >= 1
    pw = cm.pack_chunks_to_gpu_svo(lib.pack_for_gpu())
    assert len(pw.sub_chunks) > 0 and len(pw.materials) == 256
    with pytest.raises(BlokError):
        V.load_and_import_vox(tmp_path / "missing.vox", cm, lib)


@pytest.mark.skipif(not (REF / "blok/src/vox_loader.cpp").exists(), reason="reference tree not present")
def test_default_palette_equals_reference_table_read_as_text():
    text = (REF / "blok/src/vox_loader.cpp").read_text()
    body = text[text.index("DEFAULT_PALETTE[256]"):]
    body = body[body.index("{") + 1:body.index("};")]
    table = np.array([int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]{8}", body)], dtype=np.uint32)
    assert len(table) == 256
    assert np.array_equal(table, O.default_palette())
    rng = np.random.default_rng(1)
    assert np.array_equal(table, V.VoxFile.load_memory(make_vox([((2, 2, 2), random_model(rng, 2, 3))])).palette())


@pytest.mark.skipif(not (REF / "assets/models").exists(), reason="reference assets not present")
def test_reference_assets_parse_identically():
    files = sorted((REF / "assets/models").glob("*.vox"))
    assert files
    for f in files:
        data = f.read_bytes()
        pv, ov = V.VoxFile.load_file(f), O.OracleVox(data)
        assert pv.model_count() == ov.n_models() >= 1
        for i in range(pv.model_count()):
            (ps, pvox), (os_, ovox) = pv.model(i), ov.model(i)
            assert ps == os_ and np.array_equal(pvox, ovox)
        assert np.array_equal(pv.palette(), ov.palette())
        plib, olib = V.MaterialLibrary(), O.OracleMaterialLibrary()
        assert np.array_equal(pv.import_materials(plib), olib.import_vox(ov))
        assert plib.pack_for_gpu().tobytes() == olib.pack().tobytes()


@pytest.mark.gpu
def test_gpu_renders_imported_model(tmp_path, sample_file):
    """The app flow of the reference (app.cpp:105-124): loadAndImportVox -> rebuildDirtyChunks -> packChunksToGpuSvo ->
    addWorld -> frame; first hits bit-exact and shaded colour within tolerance vs the oracle."""
    from blok_amd.tracer import HipTracer
    from tests.conftest import records_equal
    path = tmp_path / "model.vox"
    path.write_bytes(sample_file)
    cm, lib = W.ChunkManager(128, 1.0), V.MaterialLibrary()
    V.load_and_import_vox(path, cm, lib)
    cm.rebuild_dirty_chunks()
    pw = cm.pack_chunks_to_gpu_svo(lib.pack_for_gpu())
    w, h = 256, 192
    tr = HipTracer(w, h).init()
    tr.add_world(pw)
    cam = W.camera_look_at((45.0, 40.0, -30.0), (10.0, 10.0, 10.0), 60.0, w, h)
    lat = O.Lattice(pw.nodes, pw.sub_chunks)
    ref, ctr = lat.trace(O.primary_rays(cam, w, h), threads=8)
    assert ctr["hits"] > 2000
    assert records_equal(tr.draw_frame(cam).reshape(-1), ref).all()
    got = tr.trace_paths(cam, spp=4, max_bounces=2, frame_index=2)
    want, _ = O.render_paths(lat, pw.materials, cam, w, h, spp=4, max_bounces=2, frame_index=2, threads=16)
    ok = (np.abs(got["color"] - want["color"]) <= 1e-4 + 1e-3 * np.abs(want["color"])).all(axis=2)
    assert ok.all(), (ok.mean(), np.argwhere(~ok)[:8].tolist())
    assert np.array_equal(got["albedo_metallic"], want["albedo_metallic"])
    tr.shutdown()
