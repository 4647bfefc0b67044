"""rtmi_write_ppm / rtmi_write_png: the files decode back to the frame that was written."""
import struct
import zlib

import numpy as np


def decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == (zlib.crc32(kind + body) & 0xffffffff)
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, color, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, color, comp, filt, lace) == (8, 6, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, w * 4 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


def test_png_and_ppm_round_trip(tmp_path):
    import raytracer_amd as rt  # host-only entry points: no device needed
    rng = np.random.default_rng(0)
    for shape in ((3, 5), (300, 257)):  # the larger one spans several stored deflate blocks
        frame = rng.integers(0, 256, size=shape + (4,), dtype=np.uint8)
        rt.write_image(tmp_path / "f.png", frame)
        assert (decode_png(tmp_path / "f.png") == frame).all()
        rt.write_image(tmp_path / "f.ppm", frame)
        blob = open(tmp_path / "f.ppm", "rb").read()
        head = f"P6\n{shape[1]} {shape[0]}\n255\n".encode()
        assert blob.startswith(head)
        assert (np.frombuffer(blob[len(head):], np.uint8).reshape(shape + (3,)) == frame[:, :, :3]).all()
