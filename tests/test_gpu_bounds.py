"""The kernels' indexing, checked on the GPU: a diagnostic build of the library (-DGSR_BOUNDS, built by
__graft_entry__.build() into gsplat.js_amd/lib_exp/bounds) counts, per site, every index derived from device data that
falls outside what it indexes -- list positions, splat indices in the lists, work-item bins and segments, LDS cells of
the binning, radix destinations.  ROCm offers no compute-sanitizer on this pool (no GPU ASan, no XNACK), so this is the
stand-in SURVEY.md section 5 names.  The frames are the parity tests' (sizes, bands, cuts, overflow and regrowth); every
counter must stay zero.  (A violation is counted, never trapped: a faulting kernel can take the node down.)"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gsplat.js_amd", "lib_exp", "bounds", "libgsplat_hip.so")
SITES = {"blend": ["work item's bin", "its segment", "list position", "splat index in a list", "item range inside the bin"],
         "bin": ["splat index of a rank", "rectangle inside the bin grid", "LDS cell of the scatter", "table row", "count cell"],
         "sort": ["destination of a radix pass", "its LDS position", "destination of the bucket sort", "key above 65536",
                  "band mode: a survivor's packed slot / original index"]}


def _counters(L):
    out = {}
    for name in SITES:
        buf = (ctypes.c_uint32 * 8)()
        assert getattr(L, "gsr_debug_bounds_" + name)(buf) == 0
        out[name] = list(buf)
    return out


def test_no_index_leaves_its_array(monkeypatch):
    import gsplat_hip as gh
    assert os.path.exists(LIB), "the bounds-checked build is missing: run python -c 'import __graft_entry__ as g; g.build()'"
    frames = 0

    def render(cfg_name, poses, size=None, n=None, seed=None, fx_div=1, **kw):
        nonlocal frames
        cfg = dict(gh.synth.CONFIGS[cfg_name])
        cfg["fx"] /= fx_div
        W, H = size or (cfg["width"], cfg["height"])
        rows = gh.synth.config_rows(cfg_name) if n is None else gh.synth.synth_rows(n, seed)
        r = gh.HIPRenderer(W, H, lib_path=LIB, **kw)
        r.set_scene_rows(rows)
        for k in poses:
            r.set_camera(gh.orbit_camera(k, 120, W, H, cfg["fx"]))
            r.render_async(); r.sync()
            frames += 1
        return r

    r = render("C1", (0, 40, 77)); L = r._L; r.dispose()
    render("C1", (3,), size=(333, 219)).dispose()                      # ragged framebuffer: partial bins and tiles
    render("C1", (5,), n=1, seed=11).dispose()
    render("C1", (5,), n=65, seed=13).dispose()
    render("C2", (13, 60)).dispose()                                   # short segments, two waves per tile
    render("C2", (13,), throughput=True).dispose()                     # one wave per tile
    render("C2", (13,), band=(864, 1056)).dispose()                    # survivor sort of a band context
    render("C3", (17,)).dispose()                                      # whole-bin work items with the saturation skip
    monkeypatch.setenv("GSR_LONG_ITEMS", "0")
    render("C3", (17,)).dispose()                                      # short segments on a dense frame: the fold of up to ~40 partials per bin
    monkeypatch.delenv("GSR_LONG_ITEMS")
    monkeypatch.setenv("GSR_SORT_ORDER", "lsd")
    render("C2", (7,)).dispose()                                       # the six-launch radix order
    render("C2", (7,), band=(0, 64)).dispose()                         # ... on a band's few survivors: most workgroups and table rows idle
    render("C2", (9,), size=(3840, 2160), band=(1728, 2208)).dispose() # ... with the rectangles carried, a 4K band
    monkeypatch.delenv("GSR_SORT_ORDER")
    render("C1", (5,), n=300, seed=17, band=(0, 32)).dispose()         # a band nothing touches: zero survivors
    render("C1", (9,), size=(3840, 2160)).dispose()                    # 8160 bins: two-level binning (cells, then chunks)
    render("C2", (9,), size=(3840, 2160)).dispose()                    # ... with cell lists of several chunks
    monkeypatch.setenv("GSR_SORT_ORDER", "lsd")
    render("C2", (31,), size=(3840, 2160)).dispose()                   # ... and the rectangles carried through the radix passes
    monkeypatch.delenv("GSR_SORT_ORDER")
    monkeypatch.setenv("GSR_BIN_TWO_LEVEL", "0")
    render("C1", (9,), size=(3840, 2160)).dispose()                    # the one-level large-grid kernels
    monkeypatch.delenv("GSR_BIN_TWO_LEVEL")
    for sub in ("1", "2"):                                             # bins of more than 64 segments: the last one takes the rest
        monkeypatch.setenv("GSR_LONG_ITEMS", "0"); monkeypatch.setenv("GSR_SEG_LEN", "256"); monkeypatch.setenv("GSR_SEG_TARGET", "100000"); monkeypatch.setenv("GSR_BLEND_SUB", sub)
        r = render("C3", (17,), size=(640, 360), fx_div=3)             # (the whole scene on 240 bins: the heaviest hold > 16384 entries)
        assert r.work_items()["seg_len"] == 256 and int(r.bin_totals().max()) > 64 * 256
        r.dispose()
    for k in ("GSR_LONG_ITEMS", "GSR_SEG_LEN", "GSR_SEG_TARGET", "GSR_BLEND_SUB"):
        monkeypatch.delenv(k)
    r = render("C1", (21,))
    r.set_list_capacity(1024)                                          # overflow: no work published, then regrowth
    r.set_camera(gh.orbit_camera(50, 120, r.width, r.height, gh.synth.CONFIGS["C1"]["fx"]))
    r.render_async(); r.sync()
    assert r.stats()["overflow_frames"] >= 1
    r.dispose()
    got = _counters(L)
    bad = {(name, SITES[name][k] if k < len(SITES[name]) else k): v for name, vals in got.items() for k, v in enumerate(vals) if v}
    assert not bad, bad
    assert frames >= 20
