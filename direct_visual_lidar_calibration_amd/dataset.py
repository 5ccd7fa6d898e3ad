"""On-disk input/output of the calibration path (SURVEY.md "next" row N3): the preprocessed-directory
format ``preprocess`` writes and ``calibrate`` reads.

======================================  ==============================================================
here                                    reference
======================================  ==============================================================
``VisualLiDARData(data_path, bag)``     ``vlcal::VisualLiDARData`` (src/vlcal/common/visual_lidar_data.cpp:10-27):
                                        ``<bag>.png`` via ``cv::imread(.., 0)`` + ``<bag>.ply`` via ``glk::load_ply``
``read_ply`` / ``write_ply``            ``glk::load_ply`` / ``glk::save_ply_binary`` (Iridescence, not in the tree);
                                        the writer emits what preprocess.cpp:161-169 hands it: float x y z + float intensity
``read_png_gray`` / ``write_png_gray``  ``cv::imread(path, 0)`` / ``cv::imwrite`` for 8-bit single-channel PNGs
``read_calib`` / ``write_calib``        ``calib.json`` (preprocess.cpp:220-232; calibrate.cpp:36-46, 56-65, 128-140)
``init_T_lidar_camera(config)``         calibrate.cpp:56-77 (manual guess first, then the automatic one)
======================================  ==============================================================

Pure host code (numpy + zlib + json); nothing here is on the per-evaluation path.
"""
import json
import os
import struct
import zlib

import numpy as np

from . import se3

# --------------------------------------------------------------------------------------------- PLY
_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4",
    "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def read_ply(path):
    """Vertex positions and intensities of a PLY file.  Returns ``(points (n, 4) float64 homogeneous,
    intensities (n,) float64 or None)`` -- ``FrameCPU(ply->vertices)`` + ``add_intensities``
    (visual_lidar_data.cpp:25-26).  ascii, binary_little_endian and binary_big_endian are accepted;
    only the vertex element is read."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    nl = data.find(b"\n", end)
    if nl < 0:
        raise ValueError(f"{path}: truncated PLY header")
    header = data[:end].decode("ascii", "replace").splitlines()
    body = nl + 1
    fmt = None
    elements = []  # (name, count, [(prop name, dtype code) | None for lists])
    for line in header[1:]:
        tok = line.split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if not elements:
                raise ValueError(f"{path}: property before any element")
            if tok[1] == "list":
                elements[-1][2].append(None)
            else:
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown PLY type {tok[1]}")
                elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    offset = body
    for name, count, props in elements:
        if name == "vertex":
            if any(p is None for p in props):
                raise ValueError(f"{path}: list property in the vertex element")
            names = [p[0] for p in props]
            if fmt == "ascii":
                text = data[offset:].split(b"\n", count)[:count]
                arr = np.array([ln.split() for ln in text], dtype=np.float64).reshape(count, len(props))
                cols = {n: arr[:, i] for i, n in enumerate(names)}
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(n, order + c) for n, c in props])
                if offset + count * dt.itemsize > len(data):
                    raise ValueError(f"{path}: truncated PLY vertex data")
                rec = np.frombuffer(data, dtype=dt, count=count, offset=offset)
                cols = {n: rec[n] for n in names}
            for k in ("x", "y", "z"):
                if k not in cols:
                    raise ValueError(f"{path}: vertex element has no '{k}' property")
            pts = np.ones((count, 4), dtype=np.float64)
            pts[:, 0], pts[:, 1], pts[:, 2] = cols["x"], cols["y"], cols["z"]
            inten = None
            for k in ("intensity", "scalar_intensity", "intensities"):
                if k in cols:
                    inten = np.ascontiguousarray(cols[k], dtype=np.float64)
                    break
            return pts, inten
        # skip an element stored before the vertices
        if fmt == "ascii":
            for _ in range(count):
                offset = data.index(b"\n", offset) + 1
        else:
            if any(p is None for p in props):
                raise ValueError(f"{path}: list element '{name}' precedes the vertices")
            offset += count * sum(np.dtype(c).itemsize for _, c in props)
    raise ValueError(f"{path}: no vertex element")


def write_ply(path, points, intensities):
    """binary_little_endian PLY with float x y z + float intensity per vertex: the record
    preprocess.cpp:161-169 fills (``p.cast<float>().head<3>()`` and the intensity narrowed to float)."""
    pts = np.asarray(points)
    n = pts.shape[0]
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    rec["intensity"] = np.asarray(intensities)
    header = f"ply\nformat binary_little_endian 1.0\nelement vertex {n}\nproperty float x\nproperty float y\nproperty float z\nproperty float intensity\nend_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())


# --------------------------------------------------------------------------------------------- PNG
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_PNG_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _unfilter(raw, height, stride, bpp):
    """Undo the per-scanline PNG filters.  None / Sub / Up rows are vectorised (Sub is a wrapping
    prefix sum per byte lane); Average / Paeth rows fall back to a scalar loop."""
    out = np.empty((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    pos = 0
    for y in range(height):
        ft = raw[pos]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=pos + 1)
        pos += stride + 1
        if ft == 0:
            cur = line.copy()
        elif ft == 1:
            cur = np.empty(stride, dtype=np.uint8)
            for c in range(bpp):
                cur[c::bpp] = np.add.accumulate(line[c::bpp], dtype=np.uint8)
        elif ft == 2:
            cur = line + prev  # uint8 wraps
        elif ft in (3, 4):
            cur = np.zeros(stride, dtype=np.int64)
            ln = line.astype(np.int64)
            up = prev.astype(np.int64)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = up[x]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = up[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (ln[x] + pred) & 255
            cur = cur.astype(np.uint8)
        else:
            raise ValueError(f"PNG: bad filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    """Decode a non-interlaced PNG (gray 1/2/4/8/16 bit, palette 1/2/4/8 bit, gray+alpha / RGB / RGBA 8/16 bit).
    Returns ``(array (H, W) or (H, W, C), bit_depth)`` with the file's own channel order (RGB)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos = 8
    ihdr = None
    idat = []
    palette = None
    while pos + 8 <= len(data):
        (length,) = struct.unpack(">I", data[pos : pos + 4])
        ctype = data[pos + 4 : pos + 8]
        chunk = data[pos + 8 : pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", chunk)
        elif ctype == b"PLTE":
            palette = np.frombuffer(chunk, dtype=np.uint8).reshape(-1, 3)
        elif ctype == b"IDAT":
            idat.append(chunk)
        elif ctype == b"IEND":
            break
    if ihdr is None or not idat:
        raise ValueError(f"{path}: PNG without IHDR / IDAT")
    width, height, depth, color, _comp, _filt, interlace = ihdr
    if interlace != 0:
        raise ValueError(f"{path}: interlaced PNGs are not supported")
    sub_byte = depth in (1, 2, 4) and color in (0, 3)
    if color not in _PNG_CHANNELS or not (depth in (8, 16) or sub_byte) or (color == 3 and depth == 16):
        raise ValueError(f"{path}: unsupported PNG colour type {color} / bit depth {depth}")
    ch = _PNG_CHANNELS[color]
    bpp = max(1, ch * depth // 8)
    stride = (width * ch * depth + 7) // 8
    raw = zlib.decompress(b"".join(idat))
    if len(raw) < height * (stride + 1):
        raise ValueError(f"{path}: truncated PNG image data")
    rows = _unfilter(raw, height, stride, bpp)
    if depth == 16:
        img = rows.reshape(height, width * ch, 2)
        img = (img[:, :, 0].astype(np.uint16) << 8) | img[:, :, 1]
    elif sub_byte:
        # packed samples, most significant bits first; gray is expanded to 8 bits by replication
        # (png_set_expand_gray_1_2_4_to_8, what cv::imread does), palette indices stay indices
        bits = np.unpackbits(rows, axis=1)[:, : width * depth].reshape(height, width, depth)
        img = np.zeros((height, width), dtype=np.uint8)
        for k in range(depth):
            img = (img << 1) | bits[:, :, k]
        if color == 0:
            img = (img.astype(np.uint16) * (255 // ((1 << depth) - 1))).astype(np.uint8)
        depth = 8
    else:
        img = rows
    img = img.reshape(height, width, ch) if ch > 1 else img.reshape(height, width)
    if color == 3:
        if palette is None:
            raise ValueError(f"{path}: palette PNG without PLTE")
        img = palette[img]
    return img, depth


def read_png_gray(path):
    """``cv::imread(path, 0)``: an 8-bit single-channel image.  16-bit samples keep their high byte;
    colour images are converted with OpenCV's fixed-point BGR2GRAY weights (R 4899, G 9617, B 1868, >> 14).
    Preprocessed directories only contain 8-bit gray PNGs, for which this is the identity."""
    img, depth = read_png(path)
    if depth == 16:
        img = (img >> 8).astype(np.uint8)
    if img.ndim == 3:
        if img.shape[2] == 2:  # gray + alpha
            img = img[:, :, 0]
        else:
            r, g, b = (img[:, :, k].astype(np.int64) for k in range(3))
            img = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    return np.ascontiguousarray(img, dtype=np.uint8)


def _png_chunk(ctype, payload):
    return struct.pack(">I", len(payload)) + ctype + payload + struct.pack(">I", zlib.crc32(ctype + payload) & 0xFFFFFFFF)


def write_png(path, image):
    """8-bit PNG, non-interlaced: (H, W) gray or (H, W, 4) RGBA (the ``_lidar_indices.png`` layout:
    the int32 index image reinterpreted as 4 bytes per pixel, preprocess.cpp:207-211).  Sub filter on
    every row, like OpenCV's encoder."""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    if img.ndim == 2:
        color, ch = 0, 1
    elif img.ndim == 3 and img.shape[2] == 4:
        color, ch = 6, 4
    elif img.ndim == 3 and img.shape[2] == 3:
        color, ch = 2, 3
    else:
        raise ValueError("write_png: (H, W), (H, W, 3) or (H, W, 4) uint8 expected")
    h, w = img.shape[:2]
    flat = img.reshape(h, w * ch)
    filt = flat.copy()
    filt[:, ch:] = flat[:, ch:] - flat[:, :-ch]  # Sub: wraps modulo 256
    raw = np.empty((h, w * ch + 1), dtype=np.uint8)
    raw[:, 0] = 1
    raw[:, 1:] = filt
    with open(path, "wb") as f:
        f.write(_PNG_SIG)
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)))
        f.write(_png_chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(_png_chunk(b"IEND", b""))


def write_png_gray(path, image):
    img = np.asarray(image)
    if img.ndim != 2:
        raise ValueError("write_png_gray: single-channel image expected")
    write_png(path, img)


# --------------------------------------------------------------------------------------------- calib.json
def read_calib(data_path):
    """``calib.json`` of a preprocessed directory.  Raises like the reference aborts (calibrate.cpp:30-34)."""
    path = os.path.join(data_path, "calib.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"error: failed to open {path}")
    with open(path) as f:
        return json.load(f)


def write_calib(data_path, config):
    """``ofs << config.dump(2)`` (nlohmann::json keeps object keys sorted)."""
    with open(os.path.join(data_path, "calib.json"), "w") as f:
        json.dump(config, f, indent=2, sort_keys=True)
        f.write("\n")


def camera_from_calib(config):
    """(camera_model, intrinsics, distortion_coeffs) -- calibrate.cpp:38-41"""
    cam = config["camera"]
    return cam["camera_model"], [float(v) for v in cam["intrinsics"]], [float(v) for v in cam["distortion_coeffs"]]


def init_T_lidar_camera(config):
    """The initial guess as TUM values [tx ty tz qx qy qz qw]: the manual one if present, else the
    automatic one, else ``None`` (the reference aborts, calibrate.cpp:56-71)."""
    res = config.get("results", {})
    for key in ("init_T_lidar_camera", "init_T_lidar_camera_auto"):
        if key in res:
            return [float(v) for v in res[key]], key
    return None, None


def tum_to_T_camera_lidar(values):
    """calibrate.cpp:73-77: T_lidar_camera from [tx ty tz qx qy qz qw] (quaternion normalised), inverted.
    Returns Sophus-order parameters [qx qy qz qw tx ty tz] of T_camera_lidar."""
    v = np.asarray(values, dtype=np.float64)
    q = v[3:7] / np.linalg.norm(v[3:7])
    T_lidar_camera = np.concatenate([q, v[0:3]])
    return se3.inverse(T_lidar_camera)


def T_camera_lidar_to_tum(x):
    """calibrate.cpp:128-133: [tx ty tz qx qy qz qw] of T_lidar_camera = inverse(T_camera_lidar)."""
    inv = se3.inverse(np.asarray(x, dtype=np.float64))
    return [float(inv[4]), float(inv[5]), float(inv[6]), float(inv[0]), float(inv[1]), float(inv[2]), float(inv[3])]


class VisualLiDARData:
    """``vlcal::VisualLiDARData``: one bag's image (8-bit gray) and cloud (points (n,4), intensities)."""

    def __init__(self, data_path, bag_name):
        png = os.path.join(data_path, bag_name + ".png")
        ply = os.path.join(data_path, bag_name + ".ply")
        if not os.path.exists(png):
            raise FileNotFoundError(f"warning: failed to load {png}")
        if not os.path.exists(ply):
            raise FileNotFoundError(f"warning: failed to load {ply}")
        self.bag_name = bag_name
        self.image = read_png_gray(png)
        self.points, self.intensities = read_ply(ply)
        if self.intensities is None:
            raise ValueError(f"{ply}: no intensity property")


def load_dataset(data_path, first_n_bags=None):
    """calibrate.cpp:36-52: ``(config, [VisualLiDARData, ...])``"""
    config = read_calib(data_path)
    names = list(config["meta"]["bag_names"])
    if first_n_bags is not None:
        names = names[: int(first_n_bags)]
    return config, [VisualLiDARData(data_path, n) for n in names]


def write_preprocessed(data_path, camera, bags, init_T_lidar_camera_tum=None, meta=None, lidar_images=None):
    """Write a directory in ``preprocess``'s output format (preprocess.cpp:150-232): ``calib.json``,
    ``<bag>.png``, ``<bag>.ply`` and, when given, ``<bag>_lidar_intensities.png`` /
    ``<bag>_lidar_indices.png``.  ``camera`` = (model, intrinsics, distortion); ``bags`` = list of
    ``(name, image_u8, points, intensities)``; ``lidar_images`` = {name: (intensity float HxW, index int32 HxW)}."""
    os.makedirs(data_path, exist_ok=True)
    for name, image, points, intensities in bags:
        write_png_gray(os.path.join(data_path, name + ".png"), image)
        write_ply(os.path.join(data_path, name + ".ply"), points, intensities)
        if lidar_images and name in lidar_images:
            inten, idx = lidar_images[name]
            # convertTo(CV_8UC1, 255.0): saturate_cast<uchar>(round-half-even(v * 255))
            write_png_gray(os.path.join(data_path, name + "_lidar_intensities.png"), np.clip(np.rint(np.asarray(inten) * 255.0), 0, 255).astype(np.uint8))
            idx4 = np.ascontiguousarray(idx, dtype="<i4").view(np.uint8).reshape(idx.shape[0], idx.shape[1], 4)
            # cv::imwrite takes the CV_8UC4 buffer as BGRA and the PNG stores RGBA: bytes 0 and 2 swap
            write_png(os.path.join(data_path, name + "_lidar_indices.png"), idx4[:, :, [2, 1, 0, 3]])
    config = {
        "meta": dict({"data_path": data_path, "bag_names": [b[0] for b in bags]}, **(meta or {})),
        "camera": {"camera_model": camera[0], "intrinsics": [float(v) for v in camera[1]], "distortion_coeffs": [float(v) for v in camera[2]]},
    }
    if init_T_lidar_camera_tum is not None:
        config["results"] = {"init_T_lidar_camera": [float(v) for v in init_T_lidar_camera_tum]}
    write_calib(data_path, config)
    return config
