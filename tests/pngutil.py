"""Minimal PNG writer for test fixtures (grey 8/16 bit or RGB 8 bit, chosen scanline filter per row)."""
import struct
import zlib

import numpy as np


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, img: np.ndarray, filters=(0,), depth=8):
    """img: [H, W] (grey) or [H, W, 3] (RGB) uint8 (or uint16 for depth=16 grey). filters: cycled over the rows."""
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    rows = img.astype(">u2").tobytes() if depth == 16 else img.astype(np.uint8).tobytes()
    bpp = ch * depth // 8
    stride = w * bpp
    raw = bytearray()
    prev = bytes(stride)
    for y in range(h):
        cur = rows[y * stride:(y + 1) * stride]
        ft = filters[y % len(filters)]
        out = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out[i] = (cur[i] - pred) & 0xFF
        raw.append(ft)
        raw += out
        prev = cur
    data = zlib.compress(bytes(raw), 6)
    # split IDAT in two chunks: readers must concatenate
    half = len(data) // 2
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)))
        f.write(_chunk(b"tEXt", b"Comment\x00rebvio test fixture"))
        f.write(_chunk(b"IDAT", data[:half]))
        f.write(_chunk(b"IDAT", data[half:]))
        f.write(_chunk(b"IEND", b""))


def write_asl(root, frames, ts_us, imu_ts_us=None, gyro=None, acc=None, shuffle_seed=None, filters=(0, 1, 2, 3, 4)):
    """EuRoC / ASL layout under `root` (= the mav0 folder): cam0/data.csv + data/<ns>.png, imu0/data.csv."""
    import os
    os.makedirs(os.path.join(root, "cam0", "data"), exist_ok=True)
    os.makedirs(os.path.join(root, "imu0"), exist_ok=True)
    order = list(range(len(frames)))
    rng = np.random.default_rng(shuffle_seed) if shuffle_seed is not None else None
    if rng is not None:
        rng.shuffle(order)
    with open(os.path.join(root, "cam0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],filename\n")
        for i in order:
            ns = int(ts_us[i]) * 1000
            f.write(f"{ns},{ns}.png\r\n")
            write_png(os.path.join(root, "cam0", "data", f"{ns}.png"), frames[i], filters=filters)
    if imu_ts_us is not None:
        idx = list(range(len(imu_ts_us)))
        if rng is not None:
            rng.shuffle(idx)
        with open(os.path.join(root, "imu0", "data.csv"), "w") as f:
            f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y,w_RS_S_z,a_RS_S_x [m s^-2],a_RS_S_y,a_RS_S_z\n")
            for k in idx:
                g, a = gyro[k], acc[k]
                f.write(f"{int(imu_ts_us[k]) * 1000},{float(g[0])!r},{float(g[1])!r},{float(g[2])!r},{float(a[0])!r},{float(a[1])!r},{float(a[2])!r}\n")
