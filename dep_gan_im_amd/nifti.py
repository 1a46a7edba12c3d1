"""Minimal NIfTI-1 single-file reader / writer (.nii, .nii.gz) on NumPy only.

The reference reads its volumes with nibabel (`nib.load(name).get_data()`, GT:97-103; nibabel is not installed
here).  What it uses of a file is the voxel array in (X, Y, Z) index order -- scaled by scl_slope / scl_inter when
the header carries them, like nibabel's get_data() -- the affine, and pixdim[4]; that is what `load` returns.
`save` writes float32 / int16 / uint8 volumes with an sform affine, for tests and for predictions
(GE saves its outputs with nib.Nifti1Image + nib.save).
"""
from __future__ import annotations

import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiError(ValueError):
    pass


class Volume:
    """image: ndarray indexed (x, y, z[, t]) -- a Fortran-ordered view of the file's voxel block, as nibabel gives;
    affine: 4x4 voxel -> world; dt: pixdim[4] (GT:103)."""

    def __init__(self, image, affine, dt, header):
        self.image, self.affine, self.dt, self.header = image, affine, dt, header


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def load(path):
    with _open(path) as f:
        raw = f.read()
    if len(raw) < 352:
        raise NiftiError("%s: shorter than a NIfTI-1 header" % path)
    for end in ("<", ">"):
        if struct.unpack(end + "i", raw[:4])[0] == 348:
            break
    else:
        raise NiftiError("%s: sizeof_hdr is not 348 (NIfTI-2 / ANALYZE are not supported)" % path)
    if raw[344:348] not in (b"n+1\x00",):
        raise NiftiError("%s: magic %r (only single-file NIfTI-1 'n+1' is supported)" % (path, raw[344:348]))
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "hh", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "hh", raw[252:256])
    if datatype not in _DTYPES:
        raise NiftiError("%s: unsupported datatype code %d" % (path, datatype))
    nd = dim[0]
    if not 1 <= nd <= 7:
        raise NiftiError("%s: dim[0] = %d" % (path, nd))
    shape = tuple(int(d) for d in dim[1:1 + nd])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    if len(raw) < off + n * dt.itemsize:
        raise NiftiError("%s: voxel block truncated (%d of %d bytes)" % (path, len(raw) - off, n * dt.itemsize))
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    if slope not in (0.0,) and np.isfinite(slope) and not (slope == 1.0 and inter == 0.0):
        arr = arr.astype(np.float64) * float(slope) + float(inter)       # nibabel's get_data() scaling
    elif dt.byteorder == ">" or (dt.byteorder == "=" and end == ">"):
        arr = arr.astype(dt.newbyteorder("="))
    if sform_code > 0:
        aff = np.eye(4)
        aff[:3, :] = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    elif qform_code > 0:
        b, c, d = struct.unpack(end + "3f", raw[256:268])
        qx, qy, qz = struct.unpack(end + "3f", raw[268:280])
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        aff = np.eye(4)
        aff[:3, :3] = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
        aff[:3, 3] = (qx, qy, qz)
    else:
        aff = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0])
    hdr = {"dim": dim, "datatype": datatype, "bitpix": bitpix, "pixdim": pixdim, "scl_slope": slope,
           "scl_inter": inter, "endianness": end}
    return Volume(arr, aff, pixdim[4], hdr)


def save(path, image, affine=None, pixdim=None):
    """Writes `image` (indexed (x, y, z[, t])) as single-file NIfTI-1; gzip when the name ends in .gz."""
    image = np.asarray(image)
    key = image.dtype.newbyteorder("=").str[1:]
    if key not in _CODES:
        raise NiftiError("unsupported dtype %s" % image.dtype)
    if not 1 <= image.ndim <= 7:
        raise NiftiError("1..7 dimensions")
    affine = np.eye(4) if affine is None else np.asarray(affine, np.float64)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [image.ndim] + list(image.shape) + [1] * (7 - image.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<hh", hdr, 70, _CODES[key], image.dtype.itemsize * 8)
    pd = [1.0] + [float(np.linalg.norm(affine[:3, i])) for i in range(3)] + [1.0] * 4
    if pixdim is not None:
        for i, v in enumerate(pixdim):
            pd[i] = float(v)
    struct.pack_into("<8f", hdr, 76, *pd)
    struct.pack_into("<3f", hdr, 108, 352.0, 0.0, 0.0)        # vox_offset; no scaling
    hdr[123] = 2 | (2 << 3)                                   # xyzt_units: mm, s
    struct.pack_into("<hh", hdr, 252, 0, 1)                   # qform none, sform scanner
    struct.pack_into("<12f", hdr, 280, *affine[:3, :].reshape(-1))
    hdr[344:348] = b"n+1\x00"
    body = bytes(hdr) + np.asfortranarray(image.astype(image.dtype.newbyteorder("<"))).tobytes(order="F")
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(body)
    else:
        with open(path, "wb") as f:
            f.write(body)
