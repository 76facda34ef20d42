"""CPU: host/png_writer.hpp (the driver's saveDir/image/<t>.png, eventCameraCalib.cpp:214-227) writes files a PNG decoder
accepts: signature, chunk CRCs, zlib stream (stored blocks, Adler-32), pixels back bit for bit — also for images whose raw
data spans several 65535-byte blocks."""
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "png_writer.hpp"
int main(int argc, char **argv) {
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<uint8_t> rgb((size_t) w * h * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            rgb[3 * ((size_t) y * w + x)] = (uint8_t) (x * 7 + y);
            rgb[3 * ((size_t) y * w + x) + 1] = (uint8_t) (x ^ y);
            rgb[3 * ((size_t) y * w + x) + 2] = (uint8_t) (y * 3);
        }
    return ecal_host::write_png_rgb(argv[1], w, h, rgb) ? 0 : 1;
}
'''


def decode_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(b):
        n, typ = struct.unpack(">I4s", b[at:at + 8])
        data = b[at + 8:at + 8 + n]
        crc, = struct.unpack(">I", b[at + 8 + n:at + 12 + n])
        assert zlib.crc32(typ + data) & 0xFFFFFFFF == crc, typ
        chunks.append((typ, data))
        at += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
    raw = zlib.decompress(chunks[1][1])          # checks the Adler-32 too
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_png_files_decode_to_the_pixels_written(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text("#include <cstdlib>\n" + SRC)
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "eventcalib_amd", "csrc", "host"), "-o", exe, str(src)])
    for w, h in ((346, 260), (1, 1), (7, 3), (300, 75)):      # 346 x 260 x 3 = 270 k bytes: five stored blocks
        out = str(tmp_path / ("a%dx%d.png" % (w, h)))
        subprocess.check_call([exe, out, str(w), str(h)])
        img = decode_png(out)
        y, x = np.mgrid[0:h, 0:w]
        want = np.stack([(x * 7 + y) & 255, (x ^ y) & 255, (y * 3) & 255], axis=2).astype(np.uint8)
        assert np.array_equal(img, want), (w, h)
