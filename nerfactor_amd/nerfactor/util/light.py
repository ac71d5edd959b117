"""Light-probe files (reference: nerfactor/models/nerfactor.py:88-93,169-179 — `xm.io.hdr.read` / `xm.io.exr.read`
followed by `imgutil.resize(new_h=light_h)`, i.e. tf.image.resize(method='bilinear', antialias=True)).

  read_hdr     Radiance RGBE (.hdr / .pic), flat and new-style run-length encoded scanlines; decoded as OpenCV does
               (the reader behind xm.io.hdr.read): channel = mantissa * 2^(exponent - 136), no +0.5 offset
  read_exr     OpenEXR needs zlib/PIZ decoders and a half-float planar layout parser; only uncompressed and ZIP /
               ZIPS scanline files with half or float R, G, B channels are supported here
  resize_antialias   separable triangle filter whose support grows with the down-scaling factor, half-pixel centres,
               weights renormalised at the borders (tf.image.resize's ScaleAndTranslate kernel)
Host-side data preparation (16 x 32 x 3 floats per probe): NumPy.
"""
import re
import struct
import zlib

import numpy as np


# ------------------------------------------------------------------------------------------------ Radiance RGBE
def read_hdr(path):
    with open(path, 'rb') as h:
        data = h.read()
    if not (data.startswith(b'#?RADIANCE') or data.startswith(b'#?RGBE')):
        raise ValueError("%s: not a Radiance HDR file" % path)
    pos, fmt = 0, None
    while True:   # header lines up to the first empty line
        end = data.index(b'\n', pos)
        line = data[pos:end].strip()
        pos = end + 1
        if not line:
            break
        if line.startswith(b'FORMAT='):
            fmt = line[7:]
    if fmt not in (None, b'32-bit_rle_rgbe'):
        raise NotImplementedError("%s: FORMAT=%s" % (path, fmt.decode()))
    end = data.index(b'\n', pos)
    m = re.fullmatch(rb'([-+])Y\s+(\d+)\s+([-+])X\s+(\d+)', data[pos:end].strip())
    if not m:
        raise NotImplementedError("%s: resolution line %r" % (path, data[pos:end]))
    height, width = int(m.group(2)), int(m.group(4))
    pos = end + 1
    buf = np.frombuffer(data, np.uint8)
    rgbe = np.empty((height, width, 4), np.uint8)
    for y in range(height):
        if 8 <= width < 32768 and buf[pos] == 2 and buf[pos + 1] == 2 and not (buf[pos + 2] & 0x80):
            if (int(buf[pos + 2]) << 8 | int(buf[pos + 3])) != width:
                raise ValueError("%s: scanline %d has the wrong width" % (path, y))
            pos += 4
            for c in range(4):   # each channel run-length encoded on its own
                x = 0
                while x < width:
                    n = int(buf[pos])
                    if n > 128:      # run
                        n -= 128
                        rgbe[y, x:x + n, c] = buf[pos + 1]
                        pos += 2
                    else:            # literal
                        rgbe[y, x:x + n, c] = buf[pos + 1:pos + 1 + n]
                        pos += 1 + n
                    if n == 0:
                        raise ValueError("%s: corrupt run-length data" % path)
                    x += n
        else:                        # flat scanline (old-style repeat pixels are not produced by modern writers)
            rgbe[y] = buf[pos:pos + 4 * width].reshape(width, 4)
            pos += 4 * width
    scale = np.ldexp(np.float32(1.), rgbe[:, :, 3].astype(np.int32) - 136).astype(np.float32)
    scale[rgbe[:, :, 3] == 0] = 0.
    img = rgbe[:, :, :3].astype(np.float32) * scale[:, :, None]
    if m.group(1) == b'+':
        img = img[::-1]
    if m.group(3) == b'-':
        img = img[:, ::-1]
    return np.ascontiguousarray(img)


def write_hdr(rgb, path):
    """Flat (uncompressed) RGBE writer — test fixtures and format conversion of probes."""
    rgb = np.asarray(rgb, np.float32)
    mx = rgb.max(-1)
    mant, exp = np.frexp(mx)
    scale = np.where(mx > 1e-32, mant * 256. / np.where(mx > 1e-32, mx, 1.), 0.).astype(np.float32)
    rgbe = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    rgbe[:, :, :3] = np.clip(rgb * scale[:, :, None], 0, 255).astype(np.uint8)
    rgbe[:, :, 3] = np.where(mx > 1e-32, exp + 128, 0).astype(np.uint8)
    with open(path, 'wb') as h:
        h.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n' % rgb.shape[:2])
        h.write(rgbe.tobytes())


# ------------------------------------------------------------------------------------------------ OpenEXR (subset)
def read_exr(path):
    with open(path, 'rb') as h:
        data = h.read()
    if struct.unpack_from('<I', data, 0)[0] != 20000630:
        raise ValueError("%s: not an OpenEXR file" % path)
    version = struct.unpack_from('<I', data, 4)[0]
    if version & 0x1E00:   # tiled / long names are fine to ignore, deep / multi-part are not
        if version & (0x800 | 0x1000 | 0x200):
            raise NotImplementedError("%s: tiled, deep or multi-part EXR" % path)
    pos, attrs = 8, {}
    while data[pos] != 0:
        end = data.index(b'\0', pos)
        name = data[pos:end].decode()
        pos = end + 1
        end = data.index(b'\0', pos)
        kind = data[pos:end].decode()
        pos = end + 1
        size = struct.unpack_from('<i', data, pos)[0]
        pos += 4
        attrs[name] = (kind, data[pos:pos + size])
        pos += size
    pos += 1
    chans, cpos, cdat = [], 0, attrs['channels'][1]
    while cdat[cpos] != 0:
        end = cdat.index(b'\0', cpos)
        ptype = struct.unpack_from('<i', cdat, end + 1)[0]
        chans.append((cdat[cpos:end].decode(), ptype))
        cpos = end + 1 + 16
    comp = attrs['compression'][1][0]
    if comp not in (0, 2, 3):
        raise NotImplementedError("%s: EXR compression %d (only NONE / ZIPS / ZIP)" % (path, comp))
    x0, y0, x1, y1 = struct.unpack('<4i', attrs['dataWindow'][1])
    width, height = x1 - x0 + 1, y1 - y0 + 1
    lines_per_block = 16 if comp == 3 else 1
    n_blocks = (height + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from('<%dQ' % n_blocks, data, pos)
    sizes = {0: 4, 1: 2, 2: 4}
    dtypes = {0: np.uint32, 1: np.float16, 2: np.float32}
    planes = {name: np.zeros((height, width), np.float32) for name, _ in chans}
    for off in offsets:
        y, nbytes = struct.unpack_from('<ii', data, off)
        raw = data[off + 8:off + 8 + nbytes]
        rows = min(lines_per_block, y1 - y + 1)
        expect = rows * width * sum(sizes[t] for _, t in chans)
        if comp and nbytes < expect:
            raw = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int32)
            raw = (np.cumsum(np.concatenate((raw[:1], raw[1:] - 128))) & 255).astype(np.uint8)   # predictor
            half = (len(raw) + 1) // 2
            out = np.empty(len(raw), np.uint8)
            out[0::2], out[1::2] = raw[:half], raw[half:]                                        # de-interleave
            raw = out.tobytes()
        p = 0
        for r in range(rows):
            for name, t in chans:   # channels are stored alphabetically, one scanline each
                n = width * sizes[t]
                planes[name][y - y0 + r] = np.frombuffer(raw, dtypes[t], width, p).astype(np.float32)
                p += n
    if not all(c in planes for c in 'RGB'):
        raise NotImplementedError("%s: channels %s" % (path, [c for c, _ in chans]))
    return np.stack([planes['R'], planes['G'], planes['B']], -1)


def read_probe(path):
    ext = path.rsplit('.', 1)[-1].lower()
    if ext == 'hdr':
        return read_hdr(path)
    if ext == 'exr':
        return read_exr(path)
    if ext == 'npy':
        return np.load(path).astype(np.float32)
    raise NotImplementedError(ext)


# ------------------------------------------------------------------------------------------------ resizing
def _triangle_weights(n_in, n_out):
    """[n_out, n_in] row-stochastic matrix of tf.image.resize(bilinear, antialias=True) along one axis."""
    scale = n_in / n_out
    support = max(scale, 1.)          # kernel radius in input pixels
    centre = (np.arange(n_out, dtype=np.float64) + 0.5) * scale
    x = np.arange(n_in, dtype=np.float64) + 0.5
    w = np.maximum(0., 1. - np.abs(x[None, :] - centre[:, None]) / support)
    return (w / w.sum(1, keepdims=True)).astype(np.float32)


def resize_antialias(img, new_h=None, new_w=None):
    """imgutil.resize of the reference (util/img.py:98-137): [H, W, C] float array, aspect-preserving when one size
    is given (the other is truncated like tf.cast)."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    if new_h is None and new_w is None:
        raise ValueError("At least one of new height or width must be given")
    if new_h is None:
        new_h = int(h / w * new_w)
    if new_w is None:
        new_w = int(w / h * new_h)
    if (new_h, new_w) == (h, w):
        return img.copy()
    out = np.einsum('ih,hwc->iwc', _triangle_weights(h, new_h), img)
    return np.einsum('jw,iwc->ijc', _triangle_weights(w, new_w), out).astype(np.float32)
