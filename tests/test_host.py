"""CPU: host-side logic of librtamd.so (no compute calls, no GPU needed).

- the shared library loads and exports every symbol include/rtamd.h declares;
- its NFF front end produces bit-identical scene arrays to the oracle's reader;
- its exact tree builder produces the same tree as the oracle's restatement of
  SpacePartition::from_primitives (reference src/space_partition.rs:197-258);
- camera basis, interleaved-tile index math, PPM writer.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import FIXTURE_DIR, NFF_DIR, REFERENCE_FIXTURES, ROOT, read_nff

EXTRA = ("extra-glass", "extra-patch", "extra-cones", "extra-onplane")


def generated(spd):
    return {
        "balls2": spd.balls(2, 64, 64),
        "tetra3": spd.tetra(3, 64, 64),
        "rings2": spd.rings(2, 64, 64),
        "lathe": spd.lathe(16, 8, 64, 64),
        "mount3": spd.mount(3, 4, 64, 64),
    }


def test_library_exports_declared_symbols(rtamd):
    header = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(rtamd.EXPORTED_SYMBOLS)
    L = C.CDLL(rtamd.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match_header(rtamd):
    assert C.sizeof(rtamd.rt_kd_node) == 64
    assert C.sizeof(rtamd.rt_view) == 52
    assert C.sizeof(rtamd.rt_surface) == 40
    assert C.sizeof(rtamd.rt_options) == 44


def scene_texts(spd):
    texts = {n: read_nff(n) for n in REFERENCE_FIXTURES + EXTRA}
    texts.update({k: v.encode() for k, v in generated(spd).items()})
    return texts


def test_nff_front_end_matches_oracle_reader(rtamd, oracle, spd):
    for name, text in scene_texts(spd).items():
        for phong in (True, False):
            p = rtamd.ParsedScene(text, use_phong=phong)
            o = oracle.OracleScene(text, use_phong=phong)
            a, d, i = p.arrays(), o.dump(), o.info
            v = p.view
            assert (v.width, v.height) == (i.width, i.height), name
            assert list(v.from_) == list(i.from_) and list(v.at) == list(i.at) and list(v.up) == list(i.up)
            assert (v.angle, v.hither) == (i.angle, i.hither)
            assert a["background"].tolist() == list(i.background)
            prims = a["primitives"]
            assert len(prims) == i.n_prims, name
            assert np.array_equal(prims["kind"].astype(np.int32), d["kind"])
            assert np.array_equal(prims["surface"].astype(np.int32), d["surface"])
            for k, pr in enumerate(prims):
                if pr["kind"] == rtamd.KIND_SPHERE:
                    s = a["spheres"][pr["index"]]
                    got = np.concatenate([s["center"], [s["radius"]], np.zeros(4, np.float32)])
                elif pr["kind"] == rtamd.KIND_CONE:
                    c = a["cones"][pr["index"]]
                    got = np.concatenate([c["base"], [c["base_radius"]], c["apex"], [c["apex_radius"]]])
                else:
                    pg = a["polygons"][pr["index"]]
                    assert pg["vertex_count"] == d["vcount"][k]
                    vs = a["vertices"][pg["first_vertex"]: pg["first_vertex"] + pg["vertex_count"]]
                    ov = d["verts6"][d["vfirst"][k]: d["vfirst"][k] + d["vcount"][k]]
                    assert np.array_equal(np.concatenate([vs["point"], vs["normal"]], axis=1).view(np.uint32),
                                          ov.view(np.uint32)), (name, k)
                    continue
                assert np.array_equal(got.astype(np.float32).view(np.uint32), d["data8"][k].view(np.uint32)), (name, k)
            s = a["surfaces"]
            got = np.stack([s["color"][:, 0], s["color"][:, 1], s["color"][:, 2], s["kd"], s["ks"], s["shine"],
                            s["reflectance"], s["transmittance"], s["ior"], s["model"].astype(np.float32)], axis=1)
            assert np.array_equal(got.astype(np.float32).view(np.uint32), d["surfaces10"].view(np.uint32)), name
            li = a["lights"]
            got = np.concatenate([li["position"], li["color"]], axis=1) if len(li) else np.zeros((0, 6), np.float32)
            assert np.array_equal(got.view(np.uint32), d["lights6"].view(np.uint32)), name


def canonical(nodes, refs):
    """Preorder (node, under subtree, over subtree) dump of the library's flat tree."""
    boxes, axis, plane, counts, out_refs = [], [], [], [], []
    stack = [0]
    while stack:
        n = nodes[stack.pop()]
        boxes.append(np.concatenate([n["bmin"], n["extent"]]))
        if n["axis"] == 3:
            axis.append(-1)
            plane.append(0.0)
            counts.append(int(n["over"]))
            out_refs.extend(refs[n["under"]: n["under"] + n["over"]].tolist())
        else:
            axis.append(int(n["axis"]))
            plane.append(float(n["plane"]))
            counts.append(0)
            stack.append(int(n["over"]))
            stack.append(int(n["under"]))
    return (np.array(boxes, np.float32), np.array(axis, np.int32), np.array(plane, np.float32),
            np.array(counts, np.uint32), np.array(out_refs, np.uint32))


def test_exact_tree_build_matches_oracle(rtamd, oracle, spd):
    some_interior = False
    for name, text in scene_texts(spd).items():
        p = rtamd.ParsedScene(text)
        info, nodes, refs = p.build_tree(threads=2)
        t = oracle.OracleScene(text).tree()
        boxes, axis, plane, counts, out_refs = canonical(nodes, refs)
        assert len(axis) == len(t["axis"]) == info.n_nodes, name
        assert np.array_equal(axis, t["axis"]), name
        assert np.array_equal(plane.view(np.uint32), t["plane"].view(np.uint32)), name
        assert np.array_equal(boxes.view(np.uint32), t["boxes"].view(np.uint32)), name
        assert np.array_equal(counts, t["count"]), name
        assert np.array_equal(out_refs, t["refs"]), name
        # bmax is the reference's re-derived max corner, fl(corner + extent)
        assert np.array_equal((nodes["bmin"] + nodes["extent"]).view(np.uint32), nodes["bmax"].view(np.uint32))
        some_interior |= info.n_nodes > 1
        oi = oracle.OracleScene(text).info
        assert (info.n_leaves, info.n_leaf_refs, info.max_depth, info.max_leaf) == (
            oi.leaves, oi.leaf_refs, oi.max_depth, oi.max_leaf)
    assert some_interior


def test_big_scene_partitions_equal_the_oracles_own(rtamd, spd):
    """tests/golden/tree_hashes.json holds SHA-256 of the partitions the CPU oracle's builder made of the
    largest benchmark scenes (minutes of CPU for the 131k-triangle one, so not rebuilt here; the GPU tests
    hold the device-costed build of that scene to its hash): the host builder on the two that take seconds."""
    import json
    from golden.make_tree_hashes import SCENES, tree_hash
    with open(os.path.join(FIXTURE_DIR, "tree_hashes.json")) as f:
        want = json.load(f)["scenes"]
    assert {"mount8_lights8", "tetra6", "rings6"} <= set(want) and want["mount8_lights8"]["primitives"] == 131076
    for key, gen, kw in SCENES:
        if key == "mount8_lights8":
            continue
        _, nodes, refs = rtamd.ParsedScene(spd.GENERATORS[gen](**kw)).build_tree(threads=4)
        assert (len(nodes), len(refs)) == (want[key]["nodes"], want[key]["leaf_refs"]), key
        assert tree_hash(*canonical(nodes, refs)) == want[key]["sha256"], key


def test_tree_build_thread_count_does_not_change_tree(rtamd, spd):
    p = rtamd.ParsedScene(spd.lathe(24, 12, 64, 64))
    _, n1, r1 = p.build_tree(threads=1)
    _, n4, r4 = p.build_tree(threads=4)
    assert n1.tobytes() == n4.tobytes() and np.array_equal(r1, r4)


def test_nonterminating_build_is_refused(rtamd):
    p = rtamd.ParsedScene(read_nff("extra-nonterminating"))
    with pytest.raises(rtamd.RtError) as e:
        p.build_tree()
    assert e.value.code == rtamd.RT_ERR_BUILD


def test_empty_scene_tree(rtamd):
    p = rtamd.ParsedScene(b"v\nfrom 0 0 0\nat 1 0 0\nup 0 1 0\nangle 45\nhither 0.1\nresolution 4 4\n")
    info, nodes, refs = p.build_tree()
    assert info.n_nodes == 1 and nodes[0]["axis"] == 3 and nodes[0]["over"] == 0 and len(refs) == 0


def test_nff_errors(rtamd):
    base = b"v\nfrom 0 0 0\nat 1 0 0\nup 0 1 0\nangle 45\nhither 0.1\nresolution 4 4\n"
    for text, msg in ((b"s 0 0 0 1\n", "missing view"), (base + b"p 2\n0 0 0\n1 0 0\n", "insufficient vertex count"),
                      (base + b"pp 3\n0 0 0 0 0 1\n1 0 0 0 0 1\n1 1 0\n", "wrong value count"),
                      (b"v\nfrom 0 0 0\n", "missing parameters"), (base + b"c\n0 0 0 1\n", "wrong value count"),
                      (base + b"s 0 0 0 0x10\n", "invalid number")):
        with pytest.raises(rtamd.RtError) as e:
            rtamd.ParsedScene(text)
        assert e.value.code == rtamd.RT_ERR_PARSE and msg in str(e.value), (text, str(e.value))
    # unknown commands and wrong argument counts are warnings, not errors (src/nff.rs:409)
    p = rtamd.ParsedScene(base + b"zz 1 2 3\nf 1 1 1\n  # indented hash is a command, not a comment\ns 0 0 0 1\n")
    assert p.warnings == 3 and p.desc.n_primitives == 1
    # column-0 comments and blank lines are skipped silently
    p = rtamd.ParsedScene(b"# c\n\n" + base)
    assert p.warnings == 0


def test_camera_basis_matches_oracle(rtamd, oracle, spd):
    for name, text in scene_texts(spd).items():
        p = rtamd.ParsedScene(text)
        o = oracle.OracleScene(text)
        for (w, h) in ((0, 0), (1920, 1080), (7, 3)):
            assert np.array_equal(p.camera_basis(w, h).view(np.uint32), o.camera(w, h).view(np.uint32)), (name, w, h)


@pytest.mark.parametrize("w,h,count", [(512, 512, 1), (512, 512, 8), (3840, 2160, 8), (30, 21, 3), (8, 8, 4), (7, 5, 2)])
def test_tile_partition_is_a_bijection(rtamd, w, h, count):
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    per_rank = [rtamd.tile_pixels(w, h, r, count) for r in range(count)]
    assert sum(per_rank) == tiles * 64
    assert max(per_rank) == rtamd.tile_pixels_max(w, h, count)
    assert max(per_rank) - min(per_rank) in (0, 64)
    xs = np.random.default_rng(0).integers(0, w, 200)
    ys = np.random.default_rng(1).integers(0, h, 200)
    seen = set()
    pts = list(zip(xs.tolist(), ys.tolist())) + [(0, 0), (w - 1, h - 1), (w - 1, 0), (0, h - 1)]
    for x, y in set(pts):
        slot, rank = rtamd.tile_slot(w, h, count, x, y)
        assert rank < count and slot < per_rank[rank]
        t = (y // 8) * ((w + 7) // 8) + x // 8
        assert rank == t % count and slot == (t // count) * 64 + (y % 8) * 8 + x % 8
        assert (rank, slot) not in seen
        seen.add((rank, slot))


def test_deinterleave_host_round_trip(rtamd):
    w, h, count = 37, 22, 3
    img = np.random.default_rng(3).integers(0, 256, (h, w, 3), dtype=np.uint8)
    stride = rtamd.tile_pixels_max(w, h, count)
    gathered = np.zeros((count, stride, 3), dtype=np.uint8)
    for y in range(h):
        for x in range(w):
            slot, rank = rtamd.tile_slot(w, h, count, x, y)
            gathered[rank, slot] = img[y, x]
    assert np.array_equal(rtamd.deinterleave_host(w, h, count, gathered), img)
    f = img.astype(np.float32) / 3
    assert np.array_equal(rtamd.deinterleave_host(w, h, count, gathered.astype(np.float32) / 3), f)


def test_ppm_writer(rtamd, tmp_path):
    img = np.random.default_rng(4).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    path = str(tmp_path / "trace.ppm")
    rtamd.write_ppm(path, img)
    data = open(path, "rb").read()
    assert data == b"P6\n7 5\n255\n" + img.tobytes()  # reference src/ppm.rs:12-25


def test_header_is_plain_c_and_host_entry_points_work_from_c(rtamd, oracle, tmp_path):
    """tests/native/abi_smoke.c: include/rtamd.h compiled as C99 with -Wall -Wextra -Werror -pedantic
    (what bindgen or a C host would see), then parse -> tree build -> camera basis -> tile math ->
    PPM writer called from C; struct sizes as C sees them equal the ctypes mirror's."""
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(rtamd.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lrtamd",
                    "-Wl,-rpath," + libdir], check=True, capture_output=True)
    nff = os.path.join(NFF_DIR, "extra-onplane.nff")
    out = str(tmp_path / "o.ppm")
    r = subprocess.run([exe, nff, out], check=True, capture_output=True, text=True)
    sizes_line, built = r.stdout.strip().splitlines()
    tok = sizes_line.split()
    sizes = {tok[i]: int(tok[i + 1]) for i in range(1, len(tok), 2)}
    for name in ("rt_view", "rt_sphere", "rt_cone", "rt_vertex", "rt_polygon", "rt_surface", "rt_light", "rt_primitive",
                 "rt_kd_node", "rt_scene_desc", "rt_options", "rt_stats", "rt_tree_info"):
        assert sizes[name] == C.sizeof(getattr(rtamd, name)), name
    kv = built.split()
    got = {kv[i]: kv[i + 1] for i in range(0, len(kv) - 3, 2) if kv[i] != "forward"}
    p = rtamd.ParsedScene(open(nff, "rb").read())
    info, nodes, refs = p.build_tree()
    assert int(got["primitives"]) == p.desc.n_primitives and int(got["nodes"]) == info.n_nodes
    assert int(got["leaf_refs"]) == info.n_leaf_refs and int(got["depth"]) == info.max_depth
    assert int(got["root_axis"]) == nodes[0]["axis"]
    assert int(got["tile_slots"]) == sum(rtamd.tile_pixels(p.view.width, p.view.height, r_, 3) for r_ in range(3))
    data = open(out, "rb").read()
    assert data.startswith(b"P6\n%d %d\n255\n" % (p.view.width, p.view.height))


def test_rust_sys_crate_mirrors_the_header(rtamd):
    """rtamd-sys/src/lib.rs (the FFI crate a Rust host links; it cannot be compiled in this image)
    declares every function of include/rtamd.h and lists every struct's fields in the header's order."""
    header = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    rs = open(os.path.join(ROOT, "rtamd-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    bound = set(re.findall(r"pub fn (rt_[a-z_0-9]+)\s*\(", rs))
    assert declared == bound, (declared - bound, bound - declared)
    plain = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for m in re.finditer(r"typedef struct (rt_[a-z_]+) \{(.*?)\} \1;", plain, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.sub(r"\[.*?\]", "", part).replace("*", " ").split()[-1])
        rm = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, flags=re.S)
        assert rm, name
        rfields = re.findall(r"pub ([a-z_0-9]+):", rm.group(1))
        assert rfields == fields, (name, fields, rfields)
    for const in re.findall(r"\b(RT_[A-Z_0-9]+) = ", plain):
        assert ("pub const %s:" % const) in rs, const
