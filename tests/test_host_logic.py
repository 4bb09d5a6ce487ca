"""CPU tests of the host side: layouts, observation descriptors, recipe tables, the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from cooking_zoo_amd import _native, soa
from cooking_zoo_amd.cooking_book.recipe_drawer import RECIPES, DEFAULT_NUM_GOALS
from cooking_zoo_amd.cooking_world.layout import feature_length
from cooking_zoo_amd.cooking_world.engine.load_level import load_meta_file
from golden_io import GoldenSet, golden_sets, layout_from_episode, recipe_table, RECIPE_NAMES
from oracle_binding import Oracle


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_feature_length_example_meta():
    assert feature_length(load_meta_file("example")) == 278
    assert feature_length(load_meta_file("large_16x16")) == 840       # SURVEY 8d cfg 5


def test_recipe_book_matches_survey_b2():
    assert DEFAULT_NUM_GOALS == 26
    ids = {n: [(x.name, x.id_num) for x in RECIPES[n]().node_list] for n in RECIPES}
    assert ids["TomatoLettuceSalad"] == [("Deliversquare", 18), ("Plate", 11), ("Tomato", 2), ("Lettuce", 0)]
    assert ids["MashedCarrotBanana"] == [("Deliversquare", 21), ("Plate", 14), ("Carrot", 9), ("Banana", 7)]
    assert ids["no_recipe"] == [("Deliversquare", 25), ("Floor", 24)]
    assert list(RECIPES) == ["TomatoSalad", "TomatoLettuceSalad", "CarrotBanana", "MashedCarrotBanana", "CucumberOnion",
                             "AppleWatermelon", "TomatoLettuceOnionSalad", "no_recipe"]
    r = RECIPES["TomatoLettuceSalad"]()
    assert r.goals_completed(26).sum() == 4 and not r.completed()
    r.set_marks(0b1111)
    assert r.goals_completed(26).sum() == 0 and r.completed()


@pytest.mark.parametrize("name", golden_sets())
def test_layout_record_and_descriptor_match_reference(name):
    """Layout.init_record reproduces the reference's freshly reset world, and the obs descriptor table,
    evaluated on the host, reproduces the reference's observation at every 7th golden step."""
    gs = GoldenSet(name)
    for ep in gs.episodes:
        lay = layout_from_episode(ep)
        rid = gs.recipe_ids
        rec = lay.init_record(ep.dims, 0, rid)
        ref = ep.states[0].copy()
        rec[soa.W_MARKS], rec[soa.W_MARKS_HI] = ref[soa.W_MARKS], ref[soa.W_MARKS_HI]
        assert np.array_equal(rec, ref), name
        desc = lay.obs_descriptor(gs.meta, ep.dims)
        for t in range(0, len(ep.states), 7):
            got = eval_descriptor(desc, ep.dims, ep.states[t])
            assert np.array_equal(bits(got), bits(ep.obs[t])), (name, t)


def eval_descriptor(desc, d, rec):
    """numpy restatement of the kernel's descriptor walk (host-side check of the table builder)."""
    lut = np.zeros(soa.LUT_SIZE)
    for i in range(2 * d.W - 1):
        lut[i] = (i - (d.W - 1)) / d.W
    for i in range(2 * d.H - 1):
        lut[soa.LUT_Y0 + i] = (i - (d.H - 1)) / d.H
    lut[soa.LUT_ONE] = 1.0
    obj0, cell0, ag0, zero = d.img_layout()
    img = np.full(zero + 2, soa.LUT_ABSENT, dtype=np.int64)
    cells = soa.record_cells(d, rec)
    ag = [soa.unpack_agent(rec[soa.AGENT_WORD0 + a]) for a in range(d.A)]
    for s_ in range(d.D):
        x, y, c, fl = soa.unpack_dyn0(rec[d.dyn0_word0 + s_])
        if fl & soa.DYN_ALIVE:
            ch, ma = bool(fl & soa.DYN_CHOPPED), bool(fl & soa.DYN_MASHED)
            img[obj0 + 6 * s_:obj0 + 6 * s_ + 6] = [x + d.W - 1, y + soa.LUT_Y0 + d.H - 1, 126 + (not (ch or ma)),
                                                                  126 + ch, 126 + ma, 127]
    for c in range(d.C):
        f = bool(cells[c] & (soa.CELL_ACTIVE | soa.CELL_WALK))
        img[cell0 + 4 * c:cell0 + 4 * c + 4] = [c % d.W + d.W - 1, c // d.W + soa.LUT_Y0 + d.H - 1, 126 + f, 127]
    for a in range(d.A):
        x, y, o, _ = ag[a]
        img[ag0 + 8 * a:ag0 + 8 * a + 7] = [x + d.W - 1, y + soa.LUT_Y0 + d.H - 1] + [126 + (o == k) for k in (1, 2, 3, 4)] + [127]
    out = np.zeros((d.A, d.F))
    for f, w in enumerate(desc):
        w = int(w)
        hw, code = (w & 0xFFFF) // 2, (w >> 16) // 4
        for a in range(d.A):
            sub = 0
            if code == soa.AX_X:
                sub = ag[a][0]
            elif code == soa.AX_Y:
                sub = ag[a][1]
            elif code >= 4 and (code - 4) // 2 != a:
                sub = ag[a][(code - 4) % 2]
            out[a, f] = lut[img[hw] - sub]
    return out


def test_c_abi_exports_every_declared_symbol(repo_root):
    """The shared library loads (no GPU needed) and exports every function include/cookingzoo.h declares."""
    hdr = open(os.path.join(repo_root, "include", "cookingzoo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cz_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cookingzoo.h but not exported"
    assert declared == {n for n, _, _ in _native.SYMBOLS}, "ctypes binding and header disagree"
    assert _native.lib().cz_abi_version() == _native.header_abi_version() >= 7


def test_struct_layouts_match_header():
    L = _native.lib()
    assert ctypes.sizeof(_native.CzConfig) == L.cz_sizeof_config() == 12 * 4 + 8 + 4 * 8
    assert ctypes.sizeof(_native.CzStats) == L.cz_sizeof_stats() == 9 * 8 + 4 * 8


def test_action_stream_matches_oracle():
    """The counter-based action stream is the same function in the library and in the oracle."""
    from oracle_binding import load_lib
    o, L = load_lib(), _native.lib()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        seed, env, ag, step = int(rng.integers(1 << 62)), int(rng.integers(1 << 40)), int(rng.integers(4)), int(rng.integers(1 << 31))
        for n in (5, 8):
            a, b = o.czo_action(seed, env, ag, step, n), L.cz_action(seed, env, ag, step, n)
            assert a == b and 0 <= a < n
    hist = np.bincount([L.cz_action(7, e, 0, t, 5) for e in range(200) for t in range(50)], minlength=5)
    assert hist.min() > 1700 and hist.max() < 2300
    o.czo_next_layout.argtypes = [ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    for e in range(50):
        for ep in range(5):
            for pool in (0, 3 | (5 << 16)):
                assert o.czo_next_layout(e, ep, pool, 11) == L.cz_next_layout(e, ep, pool, 11)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No CPU fallback: without the HIP library the product path raises (it never reaches for the oracle)."""
    import importlib
    import subprocess
    import sys
    code = ("import os, sys; os.environ['CZ_LIB'] = '/nonexistent/libcookingzoo_hip.so'; sys.path.insert(0, %r);"
            "from cooking_zoo_amd.vec_env import CookingVecEnv\n"
            "try:\n    CookingVecEnv(2, 'coop_test', 'example', 1, 10, ['TomatoSalad'], action_scheme='scheme3')\n"
            "except Exception as e:\n    print(type(e).__name__, str(e)[:60]); sys.exit(0)\nsys.exit(3)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "NativeError" in out.stdout and "not found" in out.stdout, out.stdout + out.stderr


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cooking_zoo_amd/ may mention it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cooking_zoo_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_binding" not in text and "libcz_oracle" not in text and "czo_" not in text.replace("czo_action", ""), f


def test_late_torch_guard_refuses_the_load_not_the_probe(tmp_path):
    """cooking_zoo_amd._native installs an import hook once the HIP library is loaded: a torch that bundles a ROCm runtime of
    its own must not be EXECUTED afterwards - but asking whether torch is installed keeps working, and a torch without a
    bundled runtime (or any other module) is left alone."""
    import importlib.machinery
    import sys
    from cooking_zoo_amd import _native
    g = _native._LateTorchGuard()
    assert g.find_spec("numpy") is None and g.find_spec("torch.nn") is None
    fake = tmp_path / "site"
    (fake / "torch" / "lib").mkdir(parents=True)
    (fake / "torch" / "__init__.py").write_text("x = 1\n")
    sys.path.insert(0, str(fake))
    try:
        assert g.find_spec("torch") is None                          # no runtime of its own: nothing to collide with
        (fake / "torch" / "lib" / "libamdhip64.so.7").write_bytes(b"")
        importlib.invalidate_caches()
        spec = g.find_spec("torch")
        assert spec is not None and spec.origin.endswith("__init__.py")     # availability probes get a spec
        with pytest.raises(ImportError, match="Import torch BEFORE"):
            spec.loader.exec_module(None)
    finally:
        sys.path.remove(str(fake))


def test_graft_entry_build_succeeds(repo_root):
    """The driver's build step: `make` of the HIP library (hipcc cross-compiles gfx950 without a GPU) and of the C oracle,
    then the ABI self-check.  Run in a child so that a stale library already loaded here cannot mask a failure."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); print('BUILD-OK')"],
                         cwd=repo_root, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "BUILD-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_stale_library_is_refused(repo_root, tmp_path):
    """A library whose cz_abi_version() differs from include/cookingzoo.h is refused at load."""
    import subprocess
    import sys
    src = tmp_path / "stale.c"
    src.write_text("int cz_abi_version(void) { return 1; }\n")
    so = tmp_path / "libstale.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    code = ("import os, sys; os.environ['CZ_LIB'] = %r; sys.path.insert(0, %r)\n"
            "from cooking_zoo_amd import _native\n"
            "try:\n    _native.lib()\nexcept _native.NativeError as e:\n    print('REFUSED', e); sys.exit(0)\nsys.exit(3)") % (str(so), repo_root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "REFUSED" in out.stdout and "ABI 1" in out.stdout, out.stdout + out.stderr


def test_package_without_the_repo_header_still_knows_its_abi(repo_root, tmp_path):
    """cooking_zoo_amd/ copied somewhere without include/ (a wheel, a vendored copy): the expected ABI number comes from the generated
    cooking_zoo_amd/_abi.py, which equals the header's; without either the failure is a NativeError that says what is missing
    (it used to be a FileNotFoundError at the first env creation)."""
    import shutil
    import subprocess
    import sys
    from cooking_zoo_amd import _abi, _native
    assert _abi.CZ_ABI_VERSION == _native.header_abi_version()
    shutil.copytree(os.path.join(repo_root, "cooking_zoo_amd"), tmp_path / "site" / "cooking_zoo_amd",
                    ignore=shutil.ignore_patterns("__pycache__", "*.o", "*_prof.so", "*_mark.so", "*_tl.so", "cuts", "*.s"))
    code = ("import sys; sys.path.insert(0, %r)\nfrom cooking_zoo_amd import _native\n"
            "assert not __import__('os').path.exists(_native.HEADER_PATH)\n"
            "L = _native.lib(); print('ABI', L.cz_abi_version(), _native.header_abi_version())") % str(tmp_path / "site")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and f"ABI {_abi.CZ_ABI_VERSION} {_abi.CZ_ABI_VERSION}" in out.stdout, out.stdout + out.stderr
    os.remove(tmp_path / "site" / "cooking_zoo_amd" / "_abi.py")
    code2 = ("import sys; sys.path.insert(0, %r)\nfrom cooking_zoo_amd import _native\n"
             "try:\n    _native.lib()\nexcept _native.NativeError as e:\n    print('REFUSED', e); sys.exit(0)\nsys.exit(3)") % str(tmp_path / "site")
    out = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and "REFUSED" in out.stdout and "_abi.py" in out.stdout, out.stdout + out.stderr
