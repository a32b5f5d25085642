"""NIfTI-1 volumes -> arrays in closest-canonical (RAS+) orientation, without nibabel.

The reference reads every modality and label through three nibabel calls (reference src/datasets/brats.py:84-92,
src/datasets/hecktor21.py:25-28): ``nib.load`` -> ``nib.as_closest_canonical`` -> ``get_fdata(dtype=float32)``.
nibabel is not installed on the MI355X image, so this module restates the published behaviour of those calls
(nibabel 5.x: ``Nifti1Header.get_best_affine``, ``orientations.io_orientation`` / ``apply_orientation``,
``ArrayProxy`` scaling) for single-file NIfTI-1 (``.nii`` / ``.nii.gz``).  PARITY UNPINNED: no nibabel and no
NIfTI fixture exists in the reference; tests/test_nifti.py checks this reader against files produced by
``write_nifti`` below and against hand-built headers (orientation flips / permutations, qform, scaling, endianness).

Known difference: integer data with ``scl_slope``/``scl_inter`` is scaled in float64 and then rounded to the requested
dtype; nibabel may scale in float32 when the slope is exactly representable (a difference of one float32 ulp).
"""
from __future__ import annotations

import gzip
import struct
from typing import Optional, Tuple

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiError(ValueError):
    pass


def _read_bytes(path: str) -> bytes:
    with open(path, "rb") as f:
        head = f.read(2)
        f.seek(0)
        if head == b"\x1f\x8b":
            with gzip.GzipFile(fileobj=f) as g:
                return g.read()
        return f.read()


def _quaternion_affine(b: float, c: float, d: float, qfac: float, pixdim, offset) -> np.ndarray:
    """NIfTI-1 method 2 (qform): unit quaternion (a, b, c, d), a = sqrt(1 - b^2 - c^2 - d^2) >= 0."""
    a2 = 1.0 - (b * b + c * c + d * d)
    if a2 < 1e-7:                                  # nibabel: renormalise when rounding pushed the norm above 1
        s = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d, a = b * s, c * s, d * s, 0.0
    else:
        a = np.sqrt(a2)
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]], dtype=np.float64)
    zooms = np.array(pixdim, dtype=np.float64).copy()
    zooms[2] *= qfac
    aff = np.eye(4)
    aff[:3, :3] = R * zooms
    aff[:3, 3] = offset
    return aff


def read_header(buf: bytes) -> dict:
    if len(buf) < 348:
        raise NiftiError("file shorter than a NIfTI-1 header")
    for end in ("<", ">"):
        if struct.unpack(end + "i", buf[0:4])[0] == 348:
            break
    else:
        if struct.unpack("<i", buf[0:4])[0] == 540 or struct.unpack(">i", buf[0:4])[0] == 540:
            raise NiftiError("NIfTI-2 is not supported")
        raise NiftiError("not a NIfTI-1 file (sizeof_hdr != 348)")
    magic = buf[344:348]
    if magic[:3] == b"ni1":
        raise NiftiError("two-file NIfTI (.hdr/.img) is not supported")
    if magic[:3] != b"n+1":
        raise NiftiError(f"bad NIfTI magic {magic!r}")
    dim = struct.unpack(end + "8h", buf[40:56])
    datatype, bitpix = struct.unpack(end + "2h", buf[70:74])
    pixdim = struct.unpack(end + "8f", buf[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", buf[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", buf[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + "6f", buf[256:280])
    srow = np.array(struct.unpack(end + "12f", buf[280:328]), dtype=np.float64).reshape(3, 4)
    ndim = int(dim[0])
    if not 1 <= ndim <= 7:
        raise NiftiError(f"dim[0]={ndim} out of range")
    shape = tuple(int(v) for v in dim[1:1 + ndim])
    if datatype not in _DTYPES:
        raise NiftiError(f"unsupported NIfTI datatype code {datatype}")
    # affine: sform, else qform, else the base affine (nibabel Nifti1Header.get_best_affine)
    zooms = [float(v) for v in pixdim[1:4]]
    if sform_code != 0:
        aff = np.eye(4)
        aff[:3, :] = srow
    elif qform_code != 0:
        qfac = float(pixdim[0])
        if qfac not in (-1.0, 1.0):
            qfac = 1.0
        aff = _quaternion_affine(float(qb), float(qc), float(qd), qfac, zooms, (qx, qy, qz))
    else:
        sh3 = (list(shape) + [1, 1, 1])[:3]
        aff = np.diag(zooms + [1.0])
        aff[:3, 3] = -(np.array(sh3, dtype=np.float64) - 1) / 2.0 * np.array(zooms)
        aff = np.diag([-1.0, 1.0, 1.0, 1.0]) @ aff       # NIfTI default: first axis runs right -> left
    return {"endian": end, "shape": shape, "dtype": np.dtype(_DTYPES[datatype]).newbyteorder(end), "bitpix": bitpix,
            "vox_offset": int(vox_offset) if vox_offset >= 352 else 352, "slope": float(slope), "inter": float(inter),
            "affine": aff, "zooms": tuple(zooms), "qform_code": int(qform_code), "sform_code": int(sform_code)}


def io_orientation(affine: np.ndarray) -> np.ndarray:
    """For each array axis: (closest world axis, direction).  nibabel.orientations.io_orientation."""
    affine = np.asarray(affine, dtype=np.float64)
    RZS = affine[:3, :3]
    zooms = np.sqrt(np.sum(RZS * RZS, axis=0))
    zooms[zooms == 0] = 1
    RS = RZS / zooms
    P, S, Qs = np.linalg.svd(RS, full_matrices=False)
    tol = S.max() * 3 * np.finfo(S.dtype).eps
    keep = S > tol
    R = P[:, keep] @ Qs[keep]
    ornt = np.full((3, 2), np.nan)
    for in_ax in range(3):
        col = R[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = int(np.argmax(np.abs(col)))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1 if col[out_ax] < 0 else 1
            R[out_ax, :] = 0
    return ornt


def apply_orientation(arr: np.ndarray, ornt: np.ndarray) -> np.ndarray:
    if np.any(np.isnan(ornt[:, 0])):
        raise NiftiError("degenerate affine: an array axis maps to no world axis")
    for ax, flip in enumerate(ornt[:, 1]):
        if flip == -1:
            arr = np.flip(arr, axis=ax)
    order = np.arange(arr.ndim)
    order[:3] = np.argsort(ornt[:, 0])
    return arr.transpose(order)


def load_canonical(path: str, dtype=np.float32) -> np.ndarray:
    """``nib.as_closest_canonical(nib.load(path)).get_fdata(dtype=dtype)``: array (X, Y, Z) with X -> right,
    Y -> anterior, Z -> superior."""
    arr, _ = load(path, dtype=dtype, canonical=True)
    return arr


def load(path: str, dtype=np.float32, canonical: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    buf = _read_bytes(path)
    h = read_header(buf)
    n = int(np.prod(h["shape"], dtype=np.int64))
    need = h["vox_offset"] + n * h["dtype"].itemsize
    if len(buf) < need:
        raise NiftiError(f"{path}: truncated ({len(buf)} bytes, header promises {need})")
    raw = np.frombuffer(buf, dtype=h["dtype"], count=n, offset=h["vox_offset"]).reshape(h["shape"], order="F")
    slope, inter = h["slope"], h["inter"]
    if slope == 0.0 or not np.isfinite(slope):
        slope, inter = 1.0, 0.0                        # "no scaling" (nibabel get_slope_inter)
    elif not np.isfinite(inter):
        raise NiftiError(f"{path}: invalid scl_inter {inter}")
    if slope == 1.0 and inter == 0.0:
        data = raw.astype(dtype)
    else:
        data = (raw.astype(np.float64) * slope + inter).astype(dtype)
    aff = h["affine"]
    if canonical and len(h["shape"]) >= 3:
        ornt = io_orientation(aff)
        if not np.array_equal(ornt, [[0, 1], [1, 1], [2, 1]]):
            data = apply_orientation(data, ornt)
    return np.ascontiguousarray(data), aff


def write_nifti(path: str, array: np.ndarray, affine: Optional[np.ndarray] = None, slope: float = 1.0, inter: float = 0.0,
                endian: str = "<", use_qform: Optional[Tuple[float, float, float, float]] = None) -> None:
    """Single-file NIfTI-1 writer (fixtures, and predictions if a caller wants them on disk).  ``array`` is stored as
    is (Fortran order on disk); ``affine`` goes to the sform unless ``use_qform=(b, c, d, qfac)`` is given."""
    array = np.asarray(array)
    code = _CODES.get(array.dtype.str[1:])
    if code is None:
        raise NiftiError(f"dtype {array.dtype} has no NIfTI code")
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    hdr = bytearray(352)
    e = endian
    struct.pack_into(e + "i", hdr, 0, 348)
    dim = [array.ndim] + list(array.shape) + [1] * (7 - array.ndim)
    struct.pack_into(e + "8h", hdr, 40, *dim)
    struct.pack_into(e + "2h", hdr, 70, code, array.dtype.itemsize * 8)
    zooms = np.sqrt((affine[:3, :3] ** 2).sum(0))
    pix = [1.0] + [float(v) for v in zooms] + [1.0] * 4
    if use_qform is not None:
        pix[0] = float(use_qform[3])
    struct.pack_into(e + "8f", hdr, 76, *pix)
    struct.pack_into(e + "3f", hdr, 108, 352.0, float(slope), float(inter))
    if use_qform is not None:
        struct.pack_into(e + "2h", hdr, 252, 1, 0)
        struct.pack_into(e + "6f", hdr, 256, *[float(v) for v in use_qform[:3]], *[float(v) for v in affine[:3, 3]])
    else:
        struct.pack_into(e + "2h", hdr, 252, 0, 1)
        struct.pack_into(e + "12f", hdr, 280, *[float(v) for v in affine[:3, :].reshape(-1)])
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + np.asarray(array, dtype=array.dtype.newbyteorder(e)).tobytes(order="F")
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)
