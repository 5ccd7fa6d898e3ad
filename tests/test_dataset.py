"""On-disk formats of the calibration path: PLY / PNG / calib.json round trips in the
layout `preprocess` writes and `calibrate` reads, and the `calibrate` command line on a synthetic
preprocessed directory (GPU)."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from direct_visual_lidar_calibration_amd import calibrate, dataset, se3, synth

CAM = ("plumb_bob", [210.0, 205.0, 160.0, 120.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 320, 240)


def test_ply_round_trip_and_variants(tmp_path):
    rng = np.random.default_rng(0)
    pts = np.ones((1000, 4))
    pts[:, :3] = rng.normal(size=(1000, 3)) * 10
    inten = np.floor(rng.random(1000) * 256) / 256
    p = str(tmp_path / "a.ply")
    dataset.write_ply(p, pts, inten)
    head = open(p, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 1000\nproperty float x\n")
    rp, ri = dataset.read_ply(p)
    assert rp.dtype == np.float64 and rp.shape == (1000, 4) and np.all(rp[:, 3] == 1.0)
    assert np.array_equal(rp[:, :3], pts[:, :3].astype(np.float32).astype(np.float64))  # PLY round trip = float32 rounding
    assert np.array_equal(ri, inten)  # k/256 is exact in float32
    # ascii, extra properties, a comment
    q = str(tmp_path / "b.ply")
    with open(q, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\nproperty double x\nproperty double y\nproperty double z\nproperty uchar red\nproperty float intensity\nend_header\n")
        f.write("1 2 3 255 0.5\n-1.5 0 2e1 0 0.25\n0 0 0 7 1\n")
    rp, ri = dataset.read_ply(q)
    assert np.array_equal(rp, [[1, 2, 3, 1], [-1.5, 0, 20, 1], [0, 0, 0, 1]]) and np.array_equal(ri, [0.5, 0.25, 1.0])
    # big endian, no intensity, a face element after the vertices
    r = str(tmp_path / "c.ply")
    with open(r, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
        f.write(struct.pack(">6f", 1, 2, 3, 4, 5, 6) + struct.pack(">B3i", 3, 0, 1, 1))
    rp, ri = dataset.read_ply(r)
    assert np.array_equal(rp[:, :3], [[1, 2, 3], [4, 5, 6]]) and ri is None
    with pytest.raises(ValueError):
        dataset.read_ply(__file__)
    open(str(tmp_path / "d.ply"), "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nend_header\n\0\0")
    with pytest.raises(ValueError, match="truncated"):
        dataset.read_ply(str(tmp_path / "d.ply"))


def _encode_png(img, depth, color, filters):
    """Reference PNG encoder for the test: applies the requested filter type per row (forward filters
    written from the PNG specification)."""
    h = img.shape[0]
    if depth == 16:
        b = img.astype(">u2").tobytes()
    else:
        b = img.astype(np.uint8).tobytes()
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[color]
    bpp = ch * depth // 8
    stride = len(b) // h
    rows = np.frombuffer(b, dtype=np.uint8).reshape(h, stride).astype(np.int64)
    raw = bytearray()
    prev = np.zeros(stride, dtype=np.int64)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = rows[y]
        left = np.concatenate([np.zeros(bpp, dtype=np.int64), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, dtype=np.int64), prev[:-bpp]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - left
        elif ft == 2:
            out = cur - prev
        elif ft == 3:
            out = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(ft)
        raw += (out & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    w = img.shape[1]
    data = zlib.compress(bytes(raw))
    # split the stream over two IDAT chunks, add an ancillary chunk
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 0)) + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", data[: len(data) // 2])
            + chunk(b"IDAT", data[len(data) // 2 :]) + chunk(b"IEND", b""))


def test_png_decoder_all_filters_depths_and_colour_types(tmp_path):
    rng = np.random.default_rng(1)
    gray = (rng.random((37, 53)) * 256).astype(np.uint8)
    for filters in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
        p = str(tmp_path / "g.png")
        open(p, "wb").write(_encode_png(gray, 8, 0, filters))
        assert np.array_equal(dataset.read_png_gray(p), gray), filters
    g16 = (rng.random((20, 31)) * 65536).astype(np.uint16)
    p = str(tmp_path / "g16.png")
    open(p, "wb").write(_encode_png(g16, 16, 0, [4, 3, 1]))
    img, depth = dataset.read_png(p)
    assert depth == 16 and np.array_equal(img, g16)
    assert np.array_equal(dataset.read_png_gray(p), (g16 >> 8).astype(np.uint8))
    rgb = (rng.random((19, 23, 3)) * 256).astype(np.uint8)
    p = str(tmp_path / "rgb.png")
    open(p, "wb").write(_encode_png(rgb, 8, 2, [4, 1, 3, 2]))
    img, _ = dataset.read_png(p)
    assert np.array_equal(img, rgb)
    r, g, b = (rgb[:, :, k].astype(np.int64) for k in range(3))
    assert np.array_equal(dataset.read_png_gray(p), ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))
    rgba = (rng.random((9, 11, 4)) * 256).astype(np.uint8)
    p = str(tmp_path / "rgba.png")
    open(p, "wb").write(_encode_png(rgba, 8, 6, [3, 4]))
    assert np.array_equal(dataset.read_png(p)[0], rgba)
    # our own writer (Sub filter) round-trips, gray and the 4-byte index layout
    p = str(tmp_path / "w.png")
    dataset.write_png_gray(p, gray)
    assert np.array_equal(dataset.read_png_gray(p), gray)
    dataset.write_png(p, rgba)
    assert np.array_equal(dataset.read_png(p)[0], rgba)
    big = (rng.random((1080, 1920)) * 256).astype(np.uint8)
    dataset.write_png_gray(p, big)
    assert np.array_equal(dataset.read_png_gray(p), big)
    with pytest.raises(ValueError):
        dataset.read_png(__file__)
    # packed 1 / 2 / 4-bit gray (expanded to 8 bits by replication) and a 4-bit palette image
    for depth in (1, 2, 4):
        w, h = 21, 5
        vals = (rng.random((h, w)) * (1 << depth)).astype(np.uint8)
        stride = (w * depth + 7) // 8
        bits = np.zeros((h, stride * 8), dtype=np.uint8)
        for k in range(depth):
            bits[:, k : w * depth : depth] = (vals >> (depth - 1 - k)) & 1
        packed = np.packbits(bits, axis=1)

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

        raw = b"".join(b"\0" + packed[y].tobytes() for y in range(h))
        p = str(tmp_path / f"g{depth}.png")
        open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
        assert np.array_equal(dataset.read_png_gray(p), vals * (255 // ((1 << depth) - 1)))
        if depth == 4:
            pal = (rng.random((16, 3)) * 256).astype(np.uint8)
            open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 4, 3, 0, 0, 0)) + chunk(b"PLTE", pal.tobytes()) + chunk(b"IDAT", zlib.compress(raw))
                               + chunk(b"IEND", b""))
            assert np.array_equal(dataset.read_png(p)[0], pal[vals])


def _write_synthetic_dir(path, n=15000, bags=1, seed=3, init_delta=(0.02, 0.4)):
    scenes = [synth.make_scene(CAM, num_points=n, seed=seed + k, init_delta=init_delta) for k in range(bags)]
    s0 = scenes[0]
    cfg = dataset.write_preprocessed(
        path, (s0.model, s0.intrinsics, s0.distortion), [(f"bag{k}", s.image_u8, s.points, s.intensities) for k, s in enumerate(scenes)],
        init_T_lidar_camera_tum=dataset.T_camera_lidar_to_tum(s0.T_camera_lidar_init), meta={"image_topic": "/image", "points_topic": "/points", "intensity_channel": "intensity"})
    return scenes, cfg


def test_preprocessed_directory_round_trip_and_cli_dry_run(tmp_path):
    d = str(tmp_path / "data")
    scenes, cfg = _write_synthetic_dir(d, n=5000, bags=2)
    assert sorted(os.listdir(d)) == ["bag0.ply", "bag0.png", "bag1.ply", "bag1.png", "calib.json"]
    raw = json.load(open(os.path.join(d, "calib.json")))
    assert raw["camera"]["camera_model"] == "plumb_bob" and raw["meta"]["bag_names"] == ["bag0", "bag1"] and len(raw["results"]["init_T_lidar_camera"]) == 7
    config, bags = dataset.load_dataset(d)
    assert len(bags) == 2
    for b, s in zip(bags, scenes):
        assert np.array_equal(b.image, s.image_u8)
        assert np.array_equal(b.points, s.points)          # synthetic clouds are float32-representable, like PLY-loaded ones
        assert np.array_equal(b.intensities, s.intensities)
    _, one = dataset.load_dataset(d, first_n_bags=1)
    assert len(one) == 1
    # initial guess: TUM [t, q] of T_lidar_camera -> Sophus-order T_camera_lidar
    vals, key = dataset.init_T_lidar_camera(config)
    assert key == "init_T_lidar_camera"
    x = dataset.tum_to_T_camera_lidar(vals)
    dt, dr = se3.delta_trans_rot(x, scenes[0].T_camera_lidar_init)
    assert dt < 1e-12 and dr < 1e-7
    assert np.allclose(dataset.T_camera_lidar_to_tum(x), vals, atol=1e-12)
    # automatic guess is used only when there is no manual one; neither -> None
    config["results"] = {"init_T_lidar_camera_auto": vals}
    assert dataset.init_T_lidar_camera(config)[1] == "init_T_lidar_camera_auto"
    assert dataset.init_T_lidar_camera({"camera": {}})[0] is None
    # the command line parses the reference's options and validates the directory without a GPU
    args = calibrate.build_parser().parse_args([d, "--nid_bins", "32", "--registration_type", "nid_nelder_mead", "--disable_culling", "--auto_quit", "--background", "--dry_run"])
    assert args.nid_bins == 32 and args.disable_culling and args.registration_type == "nid_nelder_mead"
    lines = []
    calibrate.run(args, log=lines.append)
    assert any("bag1" in ln for ln in lines)
    defaults = calibrate.build_parser().parse_args([d])
    assert (defaults.nid_bins, defaults.registration_type, defaults.nelder_mead_init_step, defaults.nelder_mead_convergence_criteria) == (16, "nid_bfgs", 1e-3, 1e-8)
    with pytest.raises(FileNotFoundError):
        dataset.load_dataset(str(tmp_path / "missing"))
    os.remove(os.path.join(d, "bag1.ply"))
    with pytest.raises(FileNotFoundError):
        dataset.load_dataset(d)


def test_lidar_image_files_layout(tmp_path):
    """<bag>_lidar_intensities.png / _lidar_indices.png as preprocess.cpp:203-212 writes them: the int32
    index image's four little-endian bytes travel as BGRA through cv::imwrite."""
    d = str(tmp_path / "data")
    s = synth.make_scene(CAM, num_points=2000, seed=9)
    idx = np.full((s.height, s.width), -1, dtype=np.int32)
    idx[5, 7] = 0x01020304
    idx[6, 8] = 70000
    inten = np.zeros((s.height, s.width))
    inten[5, 7] = 0.5
    inten[6, 8] = 1.0
    dataset.write_preprocessed(d, (s.model, s.intrinsics, s.distortion), [("b", s.image_u8, s.points, s.intensities)], lidar_images={"b": (inten, idx)})
    rgba, _ = dataset.read_png(os.path.join(d, "b_lidar_indices.png"))
    bgra = rgba[:, :, [2, 1, 0, 3]]  # what cv::imread(.., -1) hands back
    assert np.array_equal(np.ascontiguousarray(bgra).view("<i4").reshape(s.height, s.width), idx)
    li = dataset.read_png_gray(os.path.join(d, "b_lidar_intensities.png"))
    assert li[5, 7] == 128 and li[6, 8] == 255 and li[0, 0] == 0  # rint(127.5) = 128 (half to even)


@pytest.mark.gpu
@pytest.mark.parametrize("reg", ["nid_bfgs", "nid_nelder_mead"])
def test_calibrate_cli_end_to_end_matches_cpu_path(tmp_path, reg):
    """`calibrate <dir>` on the GPU engine writes results.T_lidar_camera within 1e-3 m / 1e-3 rad of the
    same driver running on the CPU oracle (BASELINE.json tolerance), two bags."""
    import oracle_lib
    from direct_visual_lidar_calibration_amd import calibration
    from test_calibration import OracleNIDCost, OracleNearest, oracle_cull

    d = str(tmp_path / "data")
    scenes, _ = _write_synthetic_dir(d, n=12000, bags=2, seed=21)
    args = calibrate.build_parser().parse_args([d, "--registration_type", reg, "--auto_quit", "--background"])
    config, init_x, x_gpu = calibrate.run(args, log=lambda *_: None)
    saved = json.load(open(os.path.join(d, "calib.json")))
    assert "T_lidar_camera" in saved["results"] and "init_T_lidar_camera" in saved["results"]
    x_saved = dataset.tum_to_T_camera_lidar(saved["results"]["T_lidar_camera"])
    dt, dr = se3.delta_trans_rot(x_saved, x_gpu)
    assert dt < 1e-9 and dr < 1e-7
    # CPU path: same host driver on the oracle cost objects, from the files
    _, bags = dataset.load_dataset(d)
    s = scenes[0]
    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    p = calibration.VisualCameraCalibrationParams(nid_bins=16, registration_type=reg)
    cal = calibration.VisualCameraCalibration(
        [(b.image, b.points, b.intensities) for b in bags], p, nid_cost_factory=lambda i, pt, it, b: OracleNIDCost(s, i, pt, it, b),
        nearest_cost_factory=lambda i, pt, it, b: OracleNearest(s, i, pt, it, b, max_fov), cull=oracle_cull(s))
    x_ref = cal.calibrate(init_x)
    dt, dr = se3.delta_trans_rot(x_ref, x_gpu)
    assert dt <= 1e-3 and dr <= 1e-3, (dt, dr)
