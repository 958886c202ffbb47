"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares (no compute without a GPU), parameter validation, DotDict, config, loader, aliases."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from deepmimo_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        g.build()
    lib = _native.load()
    header = open(os.path.join(ROOT, "include", "deepmimo_amd.h")).read()
    declared = set(re.findall(r"\b(dmx_[a-z_0-9]+)\s*\(", header))
    declared -= {"dmx_rays", "dmx_params", "dmx_side"}
    assert declared == set(_native.EXPORTED_SYMBOLS), (declared, _native.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.dmx_version() == _native.ABI_VERSION
    # host-only entry points are callable without a GPU
    p = _native.DmxParams()
    p.num_paths = 25
    assert lib.dmx_workspace_bytes(p, 1000, 25) >= 1000 * 25 * (3 * 4 + 4 * 8)
    assert lib.dmx_workspace_bytes(p, 0, 25) % 256 == 0
    assert np.isnan(lib.dmx_decode_max_delay(0))


def test_struct_sizes_match_header_layout():
    """ctypes mirrors must have the C layout (natural alignment, 8-byte pointers)."""
    from deepmimo_amd import _native as n
    import ctypes as C
    assert C.sizeof(n.DmxRays) == 8 + 4 + 4 + 10 * 8
    assert C.sizeof(n.DmxSide) == 10 * 8
    assert n.DmxParams.ue_rotation_per_user.offset % 8 == 0
    assert n.DmxParams.selected_subcarriers.offset % 8 == 0
    assert n.DmxParams.carrier_freq.offset % 8 == 0


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    ds = dm.Dataset(dict(onp.synth_rays(4, 3, seed=1)))
    with pytest.raises(RuntimeError, match="no GPU"):
        ds.compute_channels()
    with pytest.raises(RuntimeError):
        _ = ds.los


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under deepmimo_amd/ may import or link it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle_np|oracle/_ref|liboracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deepmimo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), (dirpath, f)


def test_channel_params_defaults_and_validate(capsys):
    import deepmimo_amd as dm
    p = dm.ChannelGenParameters()
    assert p.bs_antenna.shape.tolist() == [8, 1] and p.ue_antenna.shape.tolist() == [1, 1]
    assert p.num_paths == 25 and p.freq_domain == 1 and p.ofdm.subcarriers == 512
    assert p.ofdm.selected_subcarriers.tolist() == [0] and p.ofdm.bandwidth == 10e6 and p.ofdm.rx_filter == 0
    assert p["ofdm"]["bandwidth"] == p.ofdm.bandwidth
    p.validate(10)
    assert capsys.readouterr().out == ""
    p.bs_antenna.fov = np.array([360, 180])             # a key v4 does not read
    p.validate(10)
    assert "unnecessary" in capsys.readouterr().out
    q = dm.ChannelGenParameters()
    q.bs_antenna.rotation = np.array([[0, 1, 2]])
    with pytest.raises(AssertionError, match="BS antenna rotation"):
        q.validate(10)
    q = dm.ChannelGenParameters()
    q.ue_antenna.rotation = np.zeros((4, 3))
    with pytest.raises(AssertionError, match="UE antenna rotation"):
        q.validate(10)
    q.validate(4)
    q = dm.ChannelGenParameters()
    q.ue_antenna.rotation = np.array([[0, 10], [0, 20], [0, 30]])
    q.validate(99)
    q = dm.ChannelGenParameters()
    q.ue_antenna.radiation_pattern = "yagi"
    with pytest.raises(AssertionError, match="radiation pattern"):
        q.validate(1)
    c = p.deepcopy()
    c.bs_antenna.shape[0] = 99
    assert p.bs_antenna.shape[0] == 8 and isinstance(c, dm.ChannelGenParameters)


def test_dotdict_and_config():
    import deepmimo_amd as dm
    d = dm.DotDict({"a": 1, "b": {"c": 2}})
    assert d.a == 1 and d.b.c == 2 and d["b"]["c"] == 2 and list(d.keys()) == ["a", "b"]
    d.x = {"y": 3}
    assert isinstance(d.x, dm.DotDict) and d.to_dict()["x"] == {"y": 3}
    with pytest.raises(AttributeError):
        d.missing
    assert d.get("missing", 5) == 5
    dm.config("gpu_device_id", 3)
    assert dm.config("gpu_device_id") == 3 and dm.config.get("use_gpu") is True
    dm.config(gpu_device_id=0)
    assert dm.config.get_all()["gpu_device_id"] == 0
    dm.config.reset()
    with pytest.raises(ValueError):
        dm.config("a", 1, 2)


def test_steering_vec_matches_oracle():
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    for shape, phi, theta in (([8, 1], 0, 0), ([4, 2], 25, -10), ([3, 5], -70, 33)):
        a = dm.steering_vec(np.array(shape), phi=phi, theta=theta, spacing=0.5)
        b = onp.steering_vec(shape, phi=phi, theta=theta, spacing=0.5)
        assert a.shape == (shape[0] * shape[1], 1)
        np.testing.assert_allclose(a, b, atol=1e-14)
        assert abs(np.linalg.norm(a) - 1) < 1e-12


def test_dataset_host_logic_without_gpu():
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(6, 4, seed=4)
    ds = dm.Dataset(dict(rays))
    assert ds.n_ue == 6 and ds.pwr is ds.power and ds["toa"] is ds.delay and ds.rx_loc is ds.rx_pos
    assert ds.distance.shape == (6,) and ds.inter_int.dtype.kind == "i"
    ds.apply_fov(bs_fov=np.array([100, 90]))
    assert ds.bs_fov.tolist() == [100, 90] and ds.ue_fov.tolist() == [360, 180]
    ds["channel"] = "sentinel"
    ds["_aod_el_rot"] = "rot"
    ds.apply_fov()
    assert "channel" not in ds.keys() and ds["_aod_el_rot"] == "rot"
    ds._clear_cache_rotated_angles()
    assert "_aod_el_rot" not in ds.keys()
    with pytest.raises(KeyError):
        ds["nope"]
    p = ds.ch_params                        # lazily resolves to defaults (dataset.py:839)
    assert isinstance(p, dm.ChannelGenParameters)
    np.testing.assert_allclose(ds.tx_ori, [0, 0, 0])


def test_loader_reads_reference_layout(tmp_path):
    """core.py:186-258 on-disk format: params.json + {key}_t000_tx000_r001.mat files."""
    import json
    import scipy.io
    import deepmimo_amd as dm
    from oracle import oracle_np as onp
    rays = onp.synth_rays(12, 7, seed=8)
    folder = tmp_path / "toy_scen"
    folder.mkdir()
    params = {"version": "4.0.0a3", "rt_params": {"frequency": 3.5e9},
              "scene": {"num_scenes": 1}, "materials": {},
              "txrx_sets": {"txrx_set_0": {"id": 0, "is_tx": True, "is_rx": False, "num_points": 1, "name": "bs"},
                            "txrx_set_1": {"id": 1, "is_tx": False, "is_rx": True, "num_points": 12, "name": "ue"}}}
    (folder / "params.json").write_text(json.dumps(params))
    for k, v in rays.items():
        scipy.io.savemat(str(folder / dm.core.get_mat_filename(k, 0, 0, 1)), {k: v})
    ds = dm.load(str(folder), max_paths=5, rx_sets={1: [0, 2, 4, 6]})
    assert isinstance(ds, dm.Dataset) and ds.power.shape == (4, 5) and ds.n_ue == 4
    np.testing.assert_array_equal(ds.aoa_az, rays["aoa_az"][[0, 2, 4, 6], :5])
    assert ds.rt_params["frequency"] == 3.5e9 and ds.txrx["rx_set_id"] == 1
    with pytest.raises(ValueError):
        dm.load(str(tmp_path / "missing"))
    with pytest.raises(Exception):
        dm.load(str(folder), tx_sets=[5])


def test_mat5_parser_finds_payload(tmp_path):
    """dmx_mat5_find (host-only) on files written exactly like the reference's converter writes them
    (scipy.io.savemat, converter_utils.py:85): dims, dtype and payload offset for plain and compressed files."""
    import ctypes as C
    import scipy.io
    from deepmimo_amd import _native as n
    from deepmimo_amd.matio import find_array
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (37, 25)).astype(np.float32)
    a[5, 20:] = np.nan
    for name, arr, comp in (("power", a, False), ("power", a, True), ("inter", np.arange(12, dtype=np.int32).reshape(4, 3), False),
                            ("delay", a.astype(np.float64), False)):
        path = tmp_path / f"{name}_{int(comp)}_{arr.dtype}.mat"
        scipy.io.savemat(str(path), {name: arr}, do_compression=comp)
        raw = path.read_bytes()
        info, image = find_array(raw, name)
        assert info.ndim == 2 and (info.dims[0], info.dims[1]) == arr.shape
        assert info.elem_bytes == arr.dtype.itemsize and info.data_bytes == arr.nbytes
        payload = np.frombuffer(image, dtype=arr.dtype, count=arr.size, offset=info.data_offset)
        np.testing.assert_array_equal(payload.reshape(arr.shape[::-1]).T, arr)      # column-major on disk
    lib = n.load()
    info = n.DmxMatInfo()
    assert lib.dmx_mat5_find(C.c_char_p(raw), len(raw), b"nope", C.byref(info)) != 0
    assert b"not found" in lib.dmx_last_error()
    assert lib.dmx_mat5_find(C.c_char_p(b"x" * 200), 200, None, C.byref(info)) != 0


def test_p2m_parser_matches_reference_parser(tmp_path):
    """dmx_p2m_parse_paths (host C++) against what the REAL reference parser returned for the same synthetic
    Wireless InSite paths file (tests/golden/p2m_paths.npz, made by oracle/gen_p2m_golden.py)."""
    from deepmimo_amd.p2m import paths_parser
    z = np.load(os.path.join(ROOT, "tests", "golden", "p2m_paths.npz"), allow_pickle=False)
    path = tmp_path / "synthetic.paths.t001_01.r002.p2m"
    path.write_text(str(z["p2m_text"]))
    got = paths_parser(str(path))
    keys = [k[4:] for k in z.files if k.startswith("ref_")]
    assert set(keys) == set(got.keys())
    for k in keys:
        assert got[k].dtype == z["ref_" + k].dtype == np.float32
        np.testing.assert_array_equal(got[k], z["ref_" + k], err_msg=k)
    # a receiver with more paths than max_paths: surplus paths are skipped, the following receiver still parses
    few = paths_parser(str(path), max_paths=3)
    n = min(3, got["power"].shape[1])
    np.testing.assert_array_equal(few["power"][:, :n], got["power"][:, :n])
    bad = tmp_path / "bad.p2m"
    bad.write_text("only one line\n")
    from deepmimo_amd._native import NativeError
    with pytest.raises(NativeError):
        paths_parser(str(bad))


def test_mat5_parser_under_address_sanitizer(tmp_path):
    """MAT-v5 locator under ASan/UBSan on an intact file, truncated prefixes and clobbered header bytes."""
    import subprocess
    import scipy.io
    a = np.random.default_rng(0).uniform(size=(50, 25)).astype(np.float32)
    mat = tmp_path / "power.mat"
    scipy.io.savemat(str(mat), {"power": a})
    exe = tmp_path / "mat5_asan"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "deepmimo_amd", "csrc", "mat5_parser.cpp"),
           os.path.join(ROOT, "tests", "native", "mat5_asan_harness.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert b.returncode == 0, b.stdout[-3000:]
    r = subprocess.run([str(exe), str(mat), "power"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "mat5 asan harness ok" in r.stdout, r.stdout[-3000:]


def test_p2m_parser_under_address_sanitizer(tmp_path):
    """Host-side C++ (text parser) built with -fsanitize=address,undefined and run on the golden file, on
    truncated prefixes and on corrupted copies: malformed input must fail cleanly, never touch memory out of bounds."""
    import subprocess
    z = np.load(os.path.join(ROOT, "tests", "golden", "p2m_paths.npz"), allow_pickle=False)
    src = tmp_path / "golden.p2m"
    src.write_text(str(z["p2m_text"]))
    exe = tmp_path / "p2m_asan"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "deepmimo_amd", "csrc", "p2m_parser.cpp"),
           os.path.join(ROOT, "tests", "native", "p2m_asan_harness.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert b.returncode == 0, b.stdout[-3000:]
    r = subprocess.run([str(exe), str(src)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "p2m asan harness ok" in r.stdout, r.stdout[-3000:]


def test_c_abi_argument_validation_without_gpu():
    """Every entry point rejects malformed arguments with a status code and a message before it touches the GPU
    (error convention of include/deepmimo_amd.h); runs on the CPU-only container."""
    import ctypes as C
    from deepmimo_amd import _native as n
    lib = n.load()

    def err():
        return lib.dmx_last_error().decode()

    p = n.DmxParams()
    p.bs_shape[0], p.bs_shape[1], p.ue_shape[0], p.ue_shape[1] = 8, 1, 1, 1
    p.num_paths, p.freq_domain, p.n_subcarriers, p.n_selected, p.bandwidth = 5, 1, 64, 0, 10e6
    r = n.DmxRays()
    r.n_ue, r.n_paths, r.ld = 4, 5, 5
    buf = (C.c_char * 65536)()
    base = (C.addressof(buf) + 255) // 256 * 256
    assert lib.dmx_path_prep(None, C.byref(p), None, 0, None, None) == -1 and "rays is NULL" in err()
    assert lib.dmx_path_prep(C.byref(r), None, None, 0, None, None) == -1 and "params is NULL" in err()
    r.ld = 3
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), None, 0, None, None) == -1 and "shape" in err()
    r.ld = 5
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), None, 0, None, None) == -1 and "ray field pointer is NULL" in err()
    for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter"):
        setattr(r, k, base)                        # plausible (host) pointers: validation must stop before any launch
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), None, 0, None, None) == -4 and "workspace too small" in err()
    need = lib.dmx_workspace_bytes(C.byref(p), 4, 5)
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), C.c_void_p(base + 8), need, None, None) == -4 and "aligned" in err()
    p.bs_pattern = 7
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), C.c_void_p(base), need, None, None) == -1 and "pattern" in err()
    p.bs_pattern = 0
    p.bs_shape[0] = 0
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), C.c_void_p(base), need, None, None) == -2
    p.bs_shape[0] = 8
    p.bandwidth = 0.0
    assert lib.dmx_channels_fd(C.byref(p), C.c_void_p(base), 4, 5, 0, 4, C.c_void_p(base), 0, None) == -1 and "bandwidth" in err()
    p.bandwidth = 10e6
    assert lib.dmx_channels_fd(C.byref(p), C.c_void_p(base), 4, 5, 2, 4, C.c_void_p(base), 0, None) == -1 and "user range" in err()
    assert lib.dmx_channels_fd(C.byref(p), C.c_void_p(base), 4, 5, 0, 4, C.c_void_p(base), 99, None) == -1 and "variant" in err()
    assert lib.dmx_channels_td(C.byref(p), C.c_void_p(base), 4, 5, 0, 4, C.c_void_p(base), None) == -1 and "freq_domain" in err()
    p.rx_filter = 1
    assert lib.dmx_channels_fd(C.byref(p), C.c_void_p(base), 4, 5, 0, 4, C.c_void_p(base), 0, None) == -1
    p.n_selected = 4
    p.selected_subcarriers = base
    assert lib.dmx_channels_fd_lpf(C.byref(p), C.c_void_p(base), 4, 5, 0, 4, None, 0, C.c_void_p(base), None) == -4
    assert lib.dmx_channels_fd_beams(C.byref(p), C.c_void_p(base), 4, 5, 0, 4, None, 3, None, 0, C.c_void_p(base), None) == -1
    assert lib.dmx_pathloss(None, 1, None, None) == -1
    assert lib.dmx_mat_to_rowmajor_f32(None, 99, 4, 4, None, 4, 4, None, None) == -1 and "data type" in err()
    assert lib.dmx_p2m_count_rx(None, 0) == -1
    # zero users: every stage is a no-op success
    r.n_ue = 0
    assert lib.dmx_path_prep(C.byref(r), C.byref(p), C.c_void_p(base), 4096, None, None) == 0


def test_reference_patch_installs_on_the_real_reference():
    """deepmimo_amd.reference_patch.install(dm) on the REAL reference package (build container only: skipped where
    /root/reference is absent, e.g. on the GPU box).  With use_gpu off the reference's own path runs untouched; with
    it on and no GPU visible the call fails loudly instead of silently computing on the CPU."""
    import io
    import sys
    from contextlib import redirect_stdout, redirect_stderr
    if not os.path.isdir("/root/reference/deepmimo"):
        pytest.skip("reference not present")
    import torch
    sys.path.insert(0, "/root/reference")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    try:
        import deepmimo as dm
    finally:
        sys.path.remove("/root/reference")
    import deepmimo_amd.reference_patch as gpu
    from oracle import oracle_np as onp
    rays = onp.synth_rays(12, 5, seed=3)
    before = dm.Dataset.compute_channels
    gpu.install(dm)
    gpu.install(dm)                                                  # idempotent
    assert dm.Dataset.compute_channels is not before and dm.Dataset.compute_channels._mi355x
    ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        dm.config("use_gpu", False)
        H = ds.compute_channels(dm.ChannelGenParameters())
    ref = onp.compute_channels(rays, onp.make_params())
    assert np.array_equal(H, ref["channel"])
    if not torch.cuda.is_available():
        with redirect_stdout(io.StringIO()):
            dm.config("use_gpu", True)
        try:
            with pytest.raises(RuntimeError, match="no GPU"):
                ds.compute_channels(dm.ChannelGenParameters())
        finally:
            with redirect_stdout(io.StringIO()):
                dm.config("use_gpu", False)


def test_fd_kernel_choice_follows_the_measured_crossovers():
    """dmx_fd_kernel_choice (host-only): what variant 0 runs.  9 = one wave per user (few subcarriers - DeepMIMO's
    default is one), 2 = matrix cores (from 9 antenna pairs on), 1 = subcarrier-per-lane vector kernel (tiny panels)."""
    import ctypes as C
    from deepmimo_amd import _native as n
    lib = n.load()

    def choice(bs, ue, K, L=25):
        p = n.DmxParams()
        p.bs_shape[0], p.bs_shape[1], p.ue_shape[0], p.ue_shape[1] = bs[0], bs[1], ue[0], ue[1]
        p.num_paths, p.freq_domain, p.n_subcarriers, p.n_selected, p.bandwidth = L, 1, max(K, 1), K, 10e6
        return lib.dmx_fd_kernel_choice(C.byref(p), L)

    assert choice((8, 1), (1, 1), 1) == 9            # reference defaults: channel.py:33-63
    assert choice((8, 8), (2, 2), 512) == 2          # headline
    assert choice((8, 8), (1, 1), 8) == 9
    assert choice((8, 8), (2, 2), 16) == 2           # 4096 outputs: matrix cores
    assert choice((16, 16), (2, 2), 2) == 9          # tables need one wave per workgroup, still ahead at K <= 4
    assert choice((16, 16), (1, 1), 8) == 2
    assert choice((8, 1), (1, 1), 16) == 9 and choice((8, 1), (1, 1), 64) == 1 and choice((8, 1), (1, 1), 512) == 1
    assert choice((8, 1), (1, 1), 1024) == 2
    assert choice((3, 3), (1, 1), 64) == 2 and choice((4, 1), (1, 1), 1024) == 1
    assert choice((64, 64), (1, 1), 1) == 2          # tables of 4096 antennas do not fit the LDS of the small kernel
    assert lib.dmx_fd_kernel_choice(None, 5) == -1
