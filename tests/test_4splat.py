"""The th3cs export path: palette-index map of a volume and the `.4spl` container.

CPU: the oracle's restatement of th3cs.cu:1199-1222 against closed-form values, and the from-scratch container
writer (fluid-sims_amd/apps/tau_4splat.h) read back the way the reference's only reader does (viewer.html:67-96).
GPU: tau3d_palette_indices against the oracle, and the th3cs program end to end."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def parse_4spl(buf):
    """viewer.html:67-96, plus the words the viewer skips"""
    magic, = struct.unpack_from("<I", buf, 0)
    version = tuple(buf[4:8])
    width, height, depth, frames, psize, flags = struct.unpack_from("<6I", buf, 8)
    pal = np.frombuffer(buf, "<f4", count=psize * 12, offset=32).reshape(psize, 12)
    ioff = 32 + psize * 48
    nvox = width * height * depth
    idx = np.frombuffer(buf, np.uint8, count=nvox * frames, offset=ioff).reshape(frames, depth, height, width)
    foot = ioff + nvox * frames
    checksum, idxoffset, end = struct.unpack_from("<IQI", buf, foot)
    return dict(magic=magic, version=version, shape=(frames, depth, height, width), psize=psize, flags=flags,
                rgb=pal[:, 8:11], palette=pal, idx=idx, checksum=checksum, idxoffset=idxoffset, end=end,
                body=buf[:foot], size=foot + 16)


def test_oracle_palette_map_closed_form(oracle_built):
    o = oracle_built.Oracle3D(8, 8, 8)
    vol = np.linspace(2.0, 6.0, 4097, dtype=np.float32).reshape(1, 1, -1)
    idx, mn, mx = o.palette_indices(vol, 0.65)
    assert (mn, mx) == (2.0, 6.0) and idx.flat[0] == 0 and idx.flat[-1] == 255
    assert idx.flat[2048] == int(0.5 ** 0.65 * 255)                       # 162
    assert np.all(np.diff(idx.ravel().astype(int)) >= 0)
    t = (np.arange(4097) / 4096.0)
    want = np.clip((t ** 0.65 * 255).astype(int), 0, 255)
    bad = idx.ravel() != want                                             # float64 vs powf: only at integer crossings
    assert bad.sum() <= 4 and np.all(np.abs(idx.ravel().astype(int) - want)[bad] == 1)
    flat, mn, mx = o.palette_indices(np.full((2, 3, 4), 7.0, np.float32))
    assert np.all(flat == 0) and mn == mx == 7.0                          # range floor 1e-12: norm = 0


def test_container_writer_matches_the_viewer_layout(tmp_path):
    src = tmp_path / "w.c"
    src.write_text(r'''
#include "tau_4splat.h"
#include <stdlib.h>
int main(int argc, char **argv) {
  enum { W = 3, H = 2, D = 2, F = 2, P = 4 };
  Splat4D pal[P];
  for (int i = 0; i < P; i++) pal[i] = create_splat4D(0, 1, 0, 1, 0, 1, 0, 1, i * 0.25f, i * 0.125f, 1.0f - i * 0.25f, 1.0f);
  uint64_t idx64[W * H * D * F]; uint8_t idx8[W * H * D * F];
  for (int i = 0; i < W * H * D * F; i++) { idx64[i] = (uint64_t)((i * 7) % P); idx8[i] = (uint8_t)idx64[i]; }
  Splat4DHeader h = create_splat4DHeader(W, H, D, F, P, 0x0004);
  Splat4DVideo v = create_splat4DVideo(h, pal, idx64);
  FILE *a = fopen(argv[1], "wb"), *b = fopen(argv[2], "wb");
  int ok = write_splat4DVideo(a, &v) && write_splat4D_u8(b, &h, pal, idx8);
  fclose(a); fclose(b);
  return ok ? 0 : 1;
}
''')
    exe = tmp_path / "w"
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "fluid-sims_amd", "apps"), str(src), "-o", str(exe)],
                   check=True)
    a, b = tmp_path / "a.4spl", tmp_path / "b.4spl"
    subprocess.run([str(exe), str(a), str(b)], check=True)
    A, B = a.read_bytes(), b.read_bytes()
    assert A == B                                                          # the reference-shaped entry point and the byte one
    v = parse_4spl(A)
    assert len(A) == v["size"] == 32 + 4 * 48 + 24 + 16
    assert v["shape"] == (2, 2, 2, 3) and v["psize"] == 4 and v["flags"] == 4
    assert A[:4] == b"4SPL" and v["version"] == (1, 0, 0, 0) and struct.pack("<I", v["end"]) == b"4END"
    assert v["idxoffset"] == 32 + 4 * 48 and v["checksum"] == zlib.crc32(v["body"])
    assert np.array_equal(v["idx"].ravel(), (np.arange(24) * 7) % 4)
    assert np.allclose(v["rgb"], [[i * 0.25, i * 0.125, 1 - i * 0.25] for i in range(4)])
    assert np.allclose(v["palette"][:, :8], [0, 1] * 4) and np.allclose(v["palette"][:, 11], 1.0)


@pytest.mark.gpu
def test_palette_indices_match_the_oracle(eng, oracle_built):
    shape = (40, 32, 24)
    e = eng.Tau3D(*shape)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(20)
    sch = e.vis(0)
    got, mn, mx = e.palette_indices(0.65)
    o = oracle_built.Oracle3D(*shape)
    want, wmn, wmx = o.palette_indices(sch, 0.65)
    assert (mn, mx) == (wmn, wmx) == (float(sch.min()), float(sch.max()))
    d = np.abs(got.astype(int) - want.astype(int))
    # integer output of a float map: identical except where pow(norm) * 255 sits on an integer (device powf vs libm)
    assert d.max() <= 1 and (d != 0).mean() < 1e-3, (d.max(), (d != 0).mean())
    assert got.max() == 255 and got.min() == 0 and len(np.unique(got)) > 50
    # the map is defined on whatever vis() produced last, any gamma
    e.vis(3)
    g2, _, _ = e.palette_indices(1.0)
    w2, _, _ = o.palette_indices(e.vis(3), 1.0)
    assert np.abs(g2.astype(int) - w2.astype(int)).max() <= 1
    e.close()


def test_oracle_palette_map_vs_reference_lines(oracle_built):
    """th3cs.cu:1199-1222 itself (oracle/_ref/libref_hostmaps.so: those lines, cut and compiled with g++) against the oracle's
    restatement: same libm powf on the same host, so the indices are identical"""
    from oracle import refcpu
    if not refcpu.available_hostmaps():
        pytest.skip("oracle/_ref/libref_hostmaps.so absent (needs /root/reference to build)")
    rng = np.random.default_rng(5)
    o = oracle_built.Oracle3D(8, 8, 8)
    for shape in ((3, 5, 7), (16, 12, 9)):
        vol = (rng.random(shape, dtype=np.float32) ** 3 * 40.0).astype(np.float32)
        want = refcpu.th3cs_palette(vol, frame=1, frames=3)
        assert (want[0] == 0).all() and (want[2] == 0).all()          # written at offset f * N only (:1207)
        got, mn, mx = o.palette_indices(vol, 0.65)
        assert np.array_equal(got.astype(np.uint64), want[1])
    flat = refcpu.th3cs_palette(np.full((2, 3, 4), 7.0, np.float32))
    assert (flat == 0).all()


@pytest.mark.gpu
def test_palette_indices_vs_reference_lines(eng):
    """tau3d_palette_indices against th3cs.cu:1199-1222 itself on the engine's schlieren volume: an integer output of a float
    map — identical except where powf(norm) * 255 sits on an integer (device powf vs libm)"""
    from oracle import refcpu
    if not refcpu.available_hostmaps():
        pytest.skip("oracle/_ref/libref_hostmaps.so absent — the palette map is NOT checked against the reference")
    e = eng.Tau3D(40, 32, 24)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(20)
    sch = e.vis(0)
    got, mn, mx = e.palette_indices(0.65)
    want = refcpu.th3cs_palette(sch)[0]
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3, (int(d.max()), float((d != 0).mean()))
    e.close()


@pytest.mark.gpu
def test_th3cs_end_to_end(eng, oracle_built, tmp_path):
    if not os.path.exists(os.path.join(BIN, "th3cs")):
        subprocess.run(["make", "-C", ROOT, "th3cs"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = tmp_path / "v.4spl"
    n, frames, spf = 32, 3, 4
    r = subprocess.run([os.path.join(BIN, "th3cs"), "--n", str(n), "--frames", str(frames), "--steps-per-frame", str(spf),
                        "--out", str(out)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert "Running Hypersonic CFD for 3 frames..." in r.stdout and "Frame 3/3 processed (t=" in r.stdout and "Export Complete!" in r.stdout
    v = parse_4spl(out.read_bytes())
    assert v["shape"] == (frames, n, n, n) and v["psize"] == 256 and v["flags"] == 4
    assert v["checksum"] == zlib.crc32(v["body"]) and v["size"] == os.path.getsize(out)
    t = np.arange(256, dtype=np.float32) / np.float32(255.0)                # thermal palette, th3cs.cu:1144-1150
    rgb = np.stack([np.minimum(1, t * 2.5), np.clip(t * 2.5 - 0.5, 0, 1), np.clip(t * 2.5 - 1.5, 0, 1)], 1)
    assert np.allclose(v["rgb"], rgb, atol=1e-6)
    # the same loop through the Python binding: identical bytes (same library, same calls)
    e = eng.Tau3D(n)
    e.init(0)
    o = oracle_built.Oracle3D(n)
    st = o.init(0)
    for f in range(frames):
        e.step(spf)
        e.vis(0)
        idx, _, _ = e.palette_indices(0.65)
        assert np.array_equal(idx, v["idx"][f]), f
        # and the reference's pipeline restated on the CPU: steps, Schlieren volume, map
        st = o.run(st, spf)
        o.fill_halo_periodic(st)
        want, _, _ = o.palette_indices(o.vis(st, 0)[0], 0.65)
        d = np.abs(idx.astype(int) - want.astype(int))
        assert d.max() <= 2 and (d != 0).mean() < 0.02, (f, d.max(), (d != 0).mean())
    e.close()
