"""Host-side image file I/O edge of the hot path (reference algorithms/utils.py:11-30).

Same contract as the reference's read_img / write_img: extension dispatch, jpg ->
8-bit, tif/tiff/png -> stored depth, arrays are H x W x 3 **BGR** like cv2.imread.
Codecs are not a GPU target: OpenCV is used when it is
installed, otherwise Pillow (8-bit) plus two small codecs for the 16-bit depths
Pillow cannot hold as RGB: baseline TIFF (uncompressed strips -- what the
reference writes with IMWRITE_TIFF_COMPRESSION=1) and 16-bit PNG (zlib; all five
scan-line filters on read).  A 16-bit image never loses depth silently: JPEG,
which cannot hold it, raises.
"""
import os
import struct
import zlib

import numpy as np

try:  # pragma: no cover
    import cv2 as _cv2
except Exception:  # noqa: BLE001
    _cv2 = None


def _tiff_read_rgb16(path):
    """Baseline TIFF, uncompressed, chunky RGB, 8 or 16 bit. Returns RGB array or None."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:2] == b"II":
        e = "<"
    elif data[:2] == b"MM":
        e = ">"
    else:
        return None
    if struct.unpack(e + "H", data[2:4])[0] != 42:
        return None
    off = struct.unpack(e + "I", data[4:8])[0]
    n = struct.unpack(e + "H", data[off:off + 2])[0]
    tags = {}
    tsize = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8}
    tfmt = {1: "B", 3: "H", 4: "I"}
    for i in range(n):
        ent = data[off + 2 + 12 * i: off + 14 + 12 * i]
        tag, typ, cnt = struct.unpack(e + "HHI", ent[:8])
        if typ not in tfmt:
            continue
        nbytes = tsize[typ] * cnt
        raw = ent[8:8 + nbytes] if nbytes <= 4 else None
        if raw is None:
            p = struct.unpack(e + "I", ent[8:12])[0]
            raw = data[p:p + nbytes]
        tags[tag] = struct.unpack(e + tfmt[typ] * cnt, raw)
    w, h = tags[256][0], tags[257][0]
    bits = tags.get(258, (1,))
    comp = tags.get(259, (1,))[0]
    spp = tags.get(277, (1,))[0]
    planar = tags.get(284, (1,))[0]
    if comp != 1 or spp != 3 or planar != 1 or bits[0] not in (8, 16):
        return None
    offs, cnts = tags[273], tags[279]
    buf = b"".join(data[o:o + c] for o, c in zip(offs, cnts))
    dt = np.dtype(np.uint8) if bits[0] == 8 else np.dtype(e + "u2")
    arr = np.frombuffer(buf, dt, count=h * w * 3).reshape(h, w, 3)
    return arr.astype(arr.dtype.newbyteorder("=")) if bits[0] == 16 else arr


def _tiff_write_rgb(path, rgb):
    h, w, _ = rgb.shape
    bits = 8 * rgb.dtype.itemsize
    pix = np.ascontiguousarray(rgb).astype(rgb.dtype.newbyteorder("<")).tobytes()
    ents = []

    def ent(tag, typ, cnt, val):
        ents.append(struct.pack("<HHII", tag, typ, cnt, val))
    data_off = 8
    ifd_off = data_off + len(pix)
    ifd_off += ifd_off & 1
    nent = 10
    extra_off = ifd_off + 2 + 12 * nent + 4
    ent(256, 4, 1, w)
    ent(257, 4, 1, h)
    ent(258, 3, 3, extra_off)           # BitsPerSample -> 3 shorts
    ent(259, 3, 1, 1)                   # no compression
    ent(262, 3, 1, 2)                   # RGB
    ent(273, 4, 1, data_off)            # one strip
    ent(277, 3, 1, 3)
    ent(278, 4, 1, h)
    ent(279, 4, 1, len(pix))
    ent(284, 3, 1, 1)
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, ifd_off))
        fh.write(pix)
        if (data_off + len(pix)) & 1:
            fh.write(b"\0")
        fh.write(struct.pack("<H", nent) + b"".join(ents) + struct.pack("<I", 0))
        fh.write(struct.pack("<HHH", bits, bits, bits))


_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _png_chunks(data):
    pos = 8
    while pos + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        yield typ, data[pos + 8:pos + 8 + n]
        pos += 12 + n


def _png_unfilter(raw, h, w, bpp):
    """Undo the PNG scan-line filters (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth) of h rows of w pixels of bpp bytes.
    Pixel (r, x) depends on its left, upper and upper-left neighbours only, so every anti-diagonal is one vector step."""
    rows = np.frombuffer(raw, np.uint8, count=h * (1 + w * bpp)).reshape(h, 1 + w * bpp)
    ft = rows[:, 0].astype(np.int64)
    px = rows[:, 1:].reshape(h, w, bpp).copy()
    if not ft.any():
        return px
    if np.all((ft == 0) | (ft == 2)):            # what this module writes: cumulative sums down the columns
        return _png_up_only(px, ft)
    out = np.zeros((h + 1, w + 1, bpp), np.int64)   # one row / column of zeros in front
    out[1:, 1:] = px
    rr_all = np.arange(h)
    for d in range(h + w - 1):
        r = rr_all[max(0, d - w + 1):min(h, d + 1)]
        x = d - r
        a, b, c = out[r + 1, x], out[r, x + 1], out[r, x]     # left, up, upper-left (already reconstructed)
        f = ft[r][:, None]
        pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
        paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        pred = np.select([f == 1, f == 2, f == 3, f == 4], [a, b, (a + b) >> 1, paeth], 0)
        out[r + 1, x + 1] = (out[r + 1, x + 1] + pred) & 0xff
    return out[1:, 1:].astype(np.uint8)


def _png_up_only(px, ft):
    """rows filtered with None (0) or Up (2) only: a row is itself or itself plus the reconstructed row above"""
    out = px.copy()
    for r in range(1, out.shape[0]):
        if ft[r] == 2:
            out[r] += out[r - 1]
    return out


def _png_read16(path):
    """16-bit PNG (gray, RGB, with or without alpha, non-interlaced) -> H x W x 3 RGB uint16, or None when the file
    is not one (8-bit and palette files go through Pillow)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != _PNG_SIG:
        return None
    ihdr, idat = None, []
    for typ, body in _png_chunks(data):
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
    if ihdr is None or ihdr[2] != 16:
        return None
    w, h, _bits, ctype, _comp, _filt, interlace = ihdr
    nch = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if nch is None or interlace:
        raise RuntimeError(f"{path}: 16-bit PNG of colour type {ctype} / interlace {interlace} needs OpenCV")
    px = _png_unfilter(zlib.decompress(b"".join(idat)), h, w, 2 * nch)
    a = px.reshape(h, w, nch, 2).astype(np.uint16)
    a = (a[..., 0] << 8) | a[..., 1]
    if nch <= 2:
        return np.repeat(a[:, :, :1], 3, 2)      # gray (+ alpha dropped, as cv2.imread's 3-channel form does not apply:
    return np.ascontiguousarray(a[:, :, :3])     # IMREAD_UNCHANGED keeps alpha; the stack path takes 3 channels)


def _png_write16(path, rgb):
    """16-bit RGB PNG: big-endian samples, Up filter, zlib level 1 (cv2.imwrite's default speed setting)."""
    h, w, _ = rgb.shape
    be = np.ascontiguousarray(rgb).astype(">u2").view(np.uint8).reshape(h, w * 6)
    filt = be.copy()
    filt[1:] -= be[:-1]
    rows = np.empty((h, 1 + w * 6), np.uint8)
    rows[:, 0] = 2
    rows[0, 0] = 0
    rows[:, 1:] = filt
    rows[0, 1:] = be[0]

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)
    with open(path, "wb") as fh:
        fh.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(rows.tobytes(), 1)) + chunk(b"IEND", b""))


def read_img(file_path):
    """H x W x 3 BGR uint8/uint16 array, or None for an unsupported extension.
    RuntimeError when the file does not exist (utils.py:12-13)."""
    if not os.path.isfile(file_path):
        raise RuntimeError("File does not exist: " + file_path)
    ext = file_path.split(".")[-1]
    if ext not in ("jpeg", "jpg", "tiff", "tif", "png"):
        return None
    if _cv2 is not None:  # pragma: no cover
        return _cv2.imread(file_path) if ext in ("jpeg", "jpg") else \
            _cv2.imread(file_path, _cv2.IMREAD_UNCHANGED)
    if ext in ("tiff", "tif"):
        rgb = _tiff_read_rgb16(file_path)
        if rgb is not None:
            return np.ascontiguousarray(rgb[:, :, ::-1])
    if ext == "png":
        rgb = _png_read16(file_path)
        if rgb is not None:
            return np.ascontiguousarray(rgb[:, :, ::-1])
    from PIL import Image
    try:
        im = Image.open(file_path)
        im.load()
    except Exception:  # noqa: BLE001  cv2.imread returns None on undecodable files
        return None
    if im.mode not in ("RGB", "L"):
        im = im.convert("RGB")
    a = np.array(im)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, 2)
    return np.ascontiguousarray(a[:, :, ::-1])


def write_img(file_path, img):
    """JPEG quality 100, TIFF uncompressed, PNG default (utils.py:23-30)."""
    ext = file_path.split(".")[-1]
    if _cv2 is not None:  # pragma: no cover
        if ext in ("jpeg", "jpg"):
            _cv2.imwrite(file_path, img, [int(_cv2.IMWRITE_JPEG_QUALITY), 100])
        elif ext in ("tiff", "tif"):
            _cv2.imwrite(file_path, img, [int(_cv2.IMWRITE_TIFF_COMPRESSION), 1])
        elif ext == "png":
            _cv2.imwrite(file_path, img)
        return
    rgb = np.ascontiguousarray(img[:, :, ::-1])
    if ext in ("tiff", "tif"):
        _tiff_write_rgb(file_path, rgb)
        return
    if rgb.dtype == np.uint16:
        if ext == "png":
            _png_write16(file_path, rgb)
            return
        # JPEG holds 8 bits per sample: writing would drop depth silently
        raise ValueError(f"cannot write a 16-bit image to '{file_path}': use .tif or .png")
    from PIL import Image
    if ext in ("jpeg", "jpg"):
        Image.fromarray(rgb).save(file_path, quality=100, subsampling=0)
    elif ext == "png":
        Image.fromarray(rgb).save(file_path)


def get_img_metadata(img):
    if img is None:
        return None, None
    return img.shape[:2], img.dtype


def validate_image(img, expected_shape=None, expected_dtype=None):
    """utils.py:56-63."""
    from .errors import BitDepthError, ShapeError
    if img is None:
        raise RuntimeError("Image is None")
    shape, dtype = get_img_metadata(img)
    if expected_shape and shape[:2] != expected_shape[:2]:
        raise ShapeError(expected_shape, shape)
    if expected_dtype and dtype != expected_dtype:
        raise BitDepthError(expected_dtype, dtype)
