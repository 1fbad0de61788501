"""Builds and runs tests/c/replay_decode.c, the plain-C replay of the reference's DecodeImageJXL call sequence
(lib/extras/dec/jxl.cc:140-669), against the in-tree libjxl_amd.so."""
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_BIN = None


def binary(tmpdir="/tmp"):
    global _BIN
    if _BIN is None:
        out = os.path.join(tmpdir, "libjxl_amd_replay_%d" % os.getpid())
        libdir = os.path.join(ROOT, "libjxl_amd", "_build")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c", "replay_decode.c"), "-o", out, "-L" + libdir, "-ljxl_amd",
                        "-Wl,-rpath," + libdir], check=True)
        _BIN = out
    return _BIN


def run(jxl_bytes, tmp_path, fmt="u8", channels=3, *extra):
    src = os.path.join(str(tmp_path), "in.jxl")
    dst = os.path.join(str(tmp_path), "out.raw")
    open(src, "wb").write(jxl_bytes)
    if os.path.exists(dst):
        os.remove(dst)
    r = subprocess.run([binary(), src, dst, fmt, str(channels)] + list(extra), capture_output=True, text=True, timeout=300)
    events = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("event ")]
    pixels = open(dst, "rb").read() if os.path.exists(dst) else None
    return r.returncode, events, r.stdout, pixels


def box(kind, payload):
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def container(codestream, pieces=2, exif=b"\0\0\0\0II*\0" + b"x" * 100):
    """ISO BMFF container: signature, ftyp, an Exif box, the codestream as `pieces` jxlp boxes with an xml box between."""
    out = box(b"JXL ", b"\r\n\x87\n") + box(b"ftyp", b"jxl \0\0\0\0jxl ") + box(b"Exif", exif)
    if pieces <= 1:
        return out + box(b"jxlc", codestream)
    step = (len(codestream) + pieces - 1) // pieces
    for i in range(pieces):
        last = i == pieces - 1
        out += box(b"jxlp", struct.pack(">I", i | (0x80000000 if last else 0)) + codestream[i * step:(i + 1) * step])
        if i == 0:
            out += box(b"xml ", b"<x/>")
    return out
