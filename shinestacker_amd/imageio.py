"""Host-side image file I/O edge of the hot path (reference algorithms/utils.py:11-30).

Same contract as the reference's read_img / write_img: extension dispatch, jpg ->
8-bit, tif/tiff/png -> stored depth, arrays are H x W x 3 **BGR** like cv2.imread.
Codecs are not a GPU target: OpenCV is used when it is
installed, otherwise Pillow (8-bit) plus a small baseline-TIFF codec for 16-bit
RGB (uncompressed strips -- what the reference writes with
IMWRITE_TIFF_COMPRESSION=1).
"""
import os
import struct

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
    from PIL import Image
    if rgb.dtype == np.uint16:
        rgb = (rgb >> 8).astype(np.uint8)  # Pillow has no 16-bit RGB; TIFF keeps full depth
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
