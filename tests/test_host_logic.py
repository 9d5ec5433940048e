"""CPU-only tests of the host side: C-ABI symbol table, variable tables vs the reference checkpoints, checkpoint
manifest, class surface, loud failure without a GPU, fast-division constants, data-parallel sharding."""
import ctypes
import inspect
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shared_library_exports_every_declared_symbol():
    from mi355 import lib as milib
    protos = milib.parse_header()
    assert len(protos) >= 55
    L = milib.get()                                   # raises if the .so is missing or lacks a declared symbol
    for name in protos:
        assert hasattr(L.cdll, name), name
    assert L.mi_abi_version() == 7
    assert L.mi_vae_desc_size() == ctypes.sizeof(milib.MiVaeDesc) and L.mi_ppo_desc_size() == ctypes.sizeof(milib.MiPpoDesc)
    # every public entry point cites the reference op it replaces
    text = open(milib.HEADER).read()
    for must in ("vae/models.py:250-253", "vae/models.py:261-264", "ppo.py:218-229", "utils.py:45-50", "train.py:176-177"):
        assert must in text


def test_mlpvae_engine_layout_is_the_reference_variable_list():
    """Round 4 (csrc/mlp_engine.hip): the flat parameter buffer of the MlpVAE engine holds the reference's trainable variables (vae/models.py:287-297, TF creation order,
    `mi355.init.mlp_vae_variables`) back to back -- the two heads as one [K, 2 z] kernel -- for the reference's sizes and for odd ones; 39.5 M parameters at the defaults."""
    from mi355 import lib as milib
    from mi355.init import mlp_vae_variables
    L = milib.get()
    assert L.mi_mlpvae_desc_size() == ctypes.sizeof(milib.MiMlpVaeDesc)
    for enc, dec, tgt in (((512, 256), (256, 512), (80, 160, 3)), ((64,), (32, 48, 64), (80, 160, 1)), ((8, 16, 24, 32), (8,), (80, 160, 3))):
        d = milib.MiMlpVaeDesc()
        d.dtype, d.max_batch, d.source_size, d.target_size, d.z_dim, d.n_enc, d.n_dec = 0, 16, 38400, int(np.prod(tgt)), 64, len(enc), len(dec)
        for i, h in enumerate(enc):
            d.enc[i] = h
        for i, h in enumerate(dec):
            d.dec[i] = h
        d.loss_kind, d.with_optimizer, d.beta, d.kl_tolerance = 0, 1, 1.0, 0.0
        nt = L.mi_mlpvae_tensor_count(ctypes.byref(d))
        off, size = np.zeros(nt, np.int64), np.zeros(nt, np.int64)
        L.mi_mlpvae_param_layout(ctypes.byref(d), off.ctypes.data, size.ctypes.data, nt)
        v = mlp_vae_variables(64, (80, 160, 3), tgt, enc, dec)
        want = []
        for name, shape in v.items():
            if name.startswith(("vae/mean/", "vae/logstd_sqare/")):
                if name == "vae/mean/kernel":
                    want.append(2 * int(np.prod(shape)))
                elif name == "vae/mean/bias":
                    want.append(2 * int(shape[0]))
                continue
            want.append(int(np.prod(shape)))
        assert list(size) == want and off[0] == 0 and (off[1:] == np.cumsum(size)[:-1]).all()
        assert L.mi_mlpvae_param_floats(ctypes.byref(d)) == sum(int(np.prod(s_)) for s_ in v.values())
    assert sum(int(np.prod(s_)) for s_ in mlp_vae_variables(64, (80, 160, 3), (80, 160, 3)).values()) == L.mi_mlpvae_param_floats(ctypes.byref(_mlp_default_desc(milib)))


def _mlp_default_desc(milib):
    d = milib.MiMlpVaeDesc()
    d.dtype, d.max_batch, d.source_size, d.target_size, d.z_dim, d.n_enc, d.n_dec = 1, 512, 38400, 38400, 64, 2, 2
    d.enc[0], d.enc[1], d.dec[0], d.dec[1] = 512, 256, 256, 512
    d.loss_kind, d.with_optimizer, d.beta, d.kl_tolerance = 0, 1, 1.0, 0.0
    return d


def test_param_layout_queries_match_reference_parameter_counts():
    from mi355 import lib as milib
    L = milib.get()
    for ct, total in ((3, 2584387), (1, 2583361)):
        d = milib.MiVaeDesc(1, 512, 80, 160, 3, ct, 64, 0, 1.0, 0.0)
        off, size = np.zeros(20, np.int64), np.zeros(20, np.int64)
        L.mi_vae_param_layout(ctypes.byref(d), off.ctypes.data, size.ctypes.data, 20)
        assert size.sum() == total and (off % 8 == 0).all() and (np.diff(off) >= size[:-1]).all()
        assert L.mi_vae_param_floats(ctypes.byref(d)) >= total
        assert L.mi_vae_workspace_bytes(ctypes.byref(d)) > 0
    p = milib.MiPpoDesc(32, 67, 2, 500, 300, 0.2, 1.0, 0.01)
    off, size = np.zeros(13, np.int64), np.zeros(13, np.int64)
    L.mi_ppo_param_layout(ctypes.byref(p), off.ctypes.data, size.ctypes.data, 13)
    assert size.sum() == 369505 + 2 * 5 * 500          # + zero rows padding 67 -> 72 in the two first-layer kernels
    bad = milib.MiVaeDesc(1, 8, 20, 20, 3, 3, 64, 0, 1.0, 0.0)
    assert L.mi_vae_param_floats(ctypes.byref(bad)) == -1 and b"geometry" in L.cdll.mi_last_error()


def test_variable_tables_match_reference_checkpoints(golden_dir):
    from mi355.init import init_ppo, init_vae, ppo_variables, vae_variables
    ref = json.load(open(os.path.join(golden_dir, "ref_variables.json")))

    def trainable(tab, prefix):
        return {k: tuple(v["shape"]) for k, v in tab.items() if k.startswith(prefix) and v["dtype"] == "float32" and "Adam" not in k and "_power" not in k}
    assert dict(vae_variables(64, (80, 160, 3), (80, 160, 3))) == trainable(ref["vae_rgb"], "vae/")
    assert dict(vae_variables(64, (80, 160, 3), (80, 160, 1))) == trainable(ref["vae_seg"], "vae/")
    assert dict(ppo_variables(67, 2)) == trainable(ref["ppo_agent"], "policy/")
    v = init_vae(0, 64, (80, 160, 3), (80, 160, 3))
    assert all(np.all(a == 0) for k, a in v.items() if k.endswith("bias"))
    lim = np.sqrt(6.0 / ((3 + 32) * 16))
    assert np.abs(v["vae/encoder/conv1/kernel"]).max() <= lim and np.abs(v["vae/encoder/conv1/kernel"]).max() > 0.9 * lim
    p = init_ppo(0, 67, 2, 1.0)
    assert np.all(p["policy/action_logstd"] == 0) and p["policy/action_mean/kernel"].std() < 0.03
    # same initial values as the oracle's independent restatement (same RNG stream by construction)
    from oracle import vae_oracle as vo
    assert all(np.array_equal(v[k], a) for k, a in vo.init_vae_params(0).items())


def test_class_surface_matches_reference_signatures(tmp_path):
    import ppo
    import utils
    import vae.models as vm
    assert list(inspect.signature(vm.VAE.__init__).parameters)[:13] == [
        "self", "source_shape", "target_shape", "build_encoder_fn", "build_decoder_fn", "z_dim", "beta", "learning_rate", "lr_decay",
        "kl_tolerance", "model_dir", "loss_fn", "training"]
    sig = inspect.signature(vm.VAE.__init__).parameters
    assert sig["z_dim"].default == 512 and sig["lr_decay"].default == 0.98 and sig["learning_rate"].default == 1e-4
    for meth in ("init_session", "save", "load_latest_checkpoint", "generate_from_latent", "reconstruct", "encode", "get_step_idx",
                 "train_one_epoch", "evaluate", "decode", "train_step"):
        assert callable(getattr(vm.VAE, meth))
    psig = inspect.signature(ppo.PPO.__init__).parameters
    assert [psig[k].default for k in ("learning_rate", "lr_decay", "epsilon", "value_scale", "entropy_scale", "initial_std", "model_dir")] == \
        [3e-4, 0.998, 0.2, 0.5, 0.01, 0.4, "./"]
    for meth in ("init_session", "save", "load_latest_checkpoint", "train", "learn", "train_step", "predict", "get_episode_idx", "get_train_step_idx",
                 "get_predict_step_idx", "write_value_to_summary", "write_dict_to_summary", "write_episodic_summaries", "update_old_policy"):
        assert callable(getattr(ppo.PPO, meth))
    assert list(inspect.signature(utils.compute_gae).parameters) == ["rewards", "values", "bootstrap_values", "terminals", "gamma", "lam"]
    # constructor side effects and attributes the scripts read (vae_common.py:18-23, train.py:98,108-110)
    v = vm.ConvVAE(source_shape=np.array([80, 160, 3]), target_shape=np.array([80, 160, 1]), z_dim=64, models_dir="vae",
                   model_dir=str(tmp_path / "m"), training=False)
    assert v.z_dim == 64 and os.path.isdir(v.checkpoint_dir) and os.path.isdir(v.log_dir) and v.dirs == [v.checkpoint_dir, v.log_dir]
    assert tuple(v.encoded_shape) == (3, 8, 256)
    with pytest.raises(AssertionError):
        vm.ConvVAE(np.array([81, 160, 3]), z_dim=64, model_dir=str(tmp_path / "x"))         # decoder yields 80 rows, not 81 (vae/models.py:265)
    # MlpVAE (vae/models.py:271-299): same surface; variables in TF creation order with tf.layers.dense default names
    mv = vm.MlpVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "y"))
    assert isinstance(mv, vm.VAE) and mv.encoder_sizes == (512, 256) and mv.decoder_sizes == (256, 512)
    assert list(mv._variables.items())[:4] == [("vae/encoder/dense/kernel", (38400, 512)), ("vae/encoder/dense/bias", (512,)),
                                                ("vae/encoder/dense_1/kernel", (512, 256)), ("vae/encoder/dense_1/bias", (256,))]
    assert list(mv._variables)[4:8] == ["vae/mean/kernel", "vae/mean/bias", "vae/logstd_sqare/kernel", "vae/logstd_sqare/bias"]
    assert list(mv._variables.items())[-2:] == [("vae/decoder/dense_2/kernel", (512, 38400)), ("vae/decoder/dense_2/bias", (38400,))]
    from oracle import vae_oracle as vo
    assert dict(mv._variables) == dict(vo.mlp_vae_variable_specs(64, (80, 160, 3)))

    class Box:
        low, high, shape = np.array([-1, 0], np.float32), np.array([1, 1], np.float32), (2,)
    m = ppo.PPO(np.array([67]), Box(), model_dir=str(tmp_path / "p"))
    assert m.dirs == [m.checkpoint_dir, m.log_dir, m.video_dir] and all(os.path.isdir(d) for d in m.dirs)
    assert m.get_episode_idx() == 0 and m.get_train_step_idx() == 0 and m.get_predict_step_idx() == 0
    m.write_episodic_summaries()
    assert m.get_episode_idx() == 1                                                          # side effect of ppo.py:271-273
    m.lr_decay = 0.5
    assert m.current_learning_rate() == pytest.approx(1.5e-4)


def test_product_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ppo
    import utils
    import vae.models as vm
    v = vm.ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "m"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        v.init_session()
    with pytest.raises(RuntimeError, match="init_session"):
        v.encode(np.zeros((1, 80, 160, 3), np.float32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        utils.compute_gae([1.0], [0.5], 0.1, [False], 0.99, 0.95)

    class Box:
        low, high, shape = np.array([-1, 0], np.float32), np.array([1, 1], np.float32), (2,)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ppo.PPO(np.array([67]), Box(), model_dir=str(tmp_path / "p")).init_session()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "carla-ppo_amd")
    offenders = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if "import oracle" in txt or "from oracle" in txt or "oracle/" in txt.replace("oracle/ is test", ""):
                    offenders.append(os.path.join(dp, f))
    assert not offenders, offenders
    src = open(os.path.join(ROOT, "bench.py")).read()
    # only the cpu_baseline leg: every oracle import sits inside that one function
    body = src[src.index("def cpu_baseline"):src.index("\ndef ", src.index("def cpu_baseline") + 1)]
    assert src.count("from oracle") == body.count("from oracle") >= 1 and "import oracle" not in src.replace("from oracle import", "")


def test_checkpoint_manifest_roundtrip_and_max_to_keep(tmp_path):
    from mi355 import checkpoint as ckpt
    d = str(tmp_path / "checkpoints")
    assert ckpt.latest(d) is None
    for step in range(7):
        ckpt.save(d, step, {"vae/mean/kernel": np.full((2, 3), step, np.float32), "vae/step_idx": np.int32(step)})
    name, allp = ckpt.read_manifest(d)
    assert name == "model.ckpt-6" and allp == ["model.ckpt-%d" % i for i in range(2, 7)]       # tf.train.Saver max_to_keep=5
    assert sorted(f for f in os.listdir(d) if f.endswith(".npz")) == ["model.ckpt-%d.npz" % i for i in range(2, 7)]
    sd = ckpt.load(ckpt.latest(d))
    assert sd["vae/mean/kernel"][0, 0] == 6 and int(sd["vae/step_idx"]) == 6
    assert 'model_checkpoint_path: "model.ckpt-6"' in open(os.path.join(d, "checkpoint")).read()
    with pytest.raises(FileNotFoundError):
        ckpt.load(os.path.join(d, "model.ckpt-99"))


def test_crc32c_known_answers():
    """mi_crc32c is CRC-32C (Castagnoli): the standard check value, the running form, and the unaligned / tail paths of the slicing loop."""
    from mi355 import lib as milib
    L = milib.get()
    assert L.mi_crc32c(0, b"123456789", 9) == 0xE3069283
    assert L.mi_crc32c(0, b"", 0) == 0 and L.mi_crc32c(0, bytes(32), 32) == 0x8A9136AA       # rfc3720 B.4: 32 zero bytes
    assert L.mi_crc32c(0, bytes([0xff] * 32), 32) == 0x62A8AB43 and L.mi_crc32c(0, bytes(range(32)), 32) == 0x46DD794E
    data = bytes(np.random.RandomState(3).randint(0, 256, 1000).astype(np.uint8))
    for cut in (0, 1, 7, 8, 9, 500, 999, 1000):
        assert L.mi_crc32c(L.mi_crc32c(0, data[:cut], cut), data[cut:], len(data) - cut) == L.mi_crc32c(0, data, len(data))


def test_tf_bundle_writer_reproduces_the_reference_index_files(golden_dir, tmp_path):
    """The reference ships the .index half of three tf.train.Saver checkpoints (rgb VAE, seg VAE, PPO agent).  Reading one verifies every
    block trailer TensorFlow wrote (masked CRC-32C) with OUR checksum code; re-writing the parsed entries must give the same bytes:
    block layout, prefix compression, restart arrays, shortened index keys, footer, BundleEntry / BundleHeader encoding."""
    from mi355 import tf_bundle as tb
    ref = json.load(open(os.path.join(golden_dir, "ref_variables.json")))
    for name in ("vae_rgb", "vae_seg", "ppo_agent"):
        path = os.path.join(golden_dir, "ref_index", name + ".index")
        entries, shards = tb.read_index(path, verify=True)
        assert shards == 1 and set(entries) == {k for k in ref[name]}
        for k, e in entries.items():
            assert list(e["shape"]) == ref[name][k]["shape"] and e["crc32c"] is not None and not e["sliced"]
        offs = sorted((e["offset"], e["size"]) for e in entries.values())
        assert offs[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(offs, offs[1:]))      # tensors back to back in the data shard
        items = [(b"", tb._header_proto(shards))] + [(k.encode(), tb._entry_proto(e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"]))
                                                      for k, e in sorted(entries.items(), key=lambda kv: kv[0].encode())]
        out = str(tmp_path / (name + ".index"))
        tb._write_table(out, items)
        assert open(out, "rb").read() == open(path, "rb").read()


def test_tf_bundle_roundtrip_and_checkpoint_dir_in_reference_format(tmp_path):
    """write_bundle -> read_bundle keeps names, dtypes, shapes (scalars too) and values over several table blocks; a flipped data byte is
    caught by the tensor checksum; checkpoint.save(fmt='tf') writes what checkpoint.load / latest restore (a directory a tf.train.Saver wrote)."""
    from mi355 import checkpoint as ckpt, tf_bundle as tb
    from mi355.init import init_vae
    rng = np.random.RandomState(0)
    v = dict(init_vae(0, 64, (80, 160, 3), (80, 160, 3)))
    v.update({k + "/Adam": rng.randn(*a.shape).astype(np.float32) for k, a in list(v.items())})
    v.update({"vae/beta1_power": np.float32(0.81), "vae/step_idx": np.array(7, np.int32), "global_step": np.array(123, np.int64)})
    for i in range(300):
        v["pad/var_%03d" % i] = rng.randn(3, 2).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-7")
    tb.write_bundle(prefix, v)
    assert sorted(os.listdir(tmp_path)) == ["model.ckpt-7.data-00000-of-00001", "model.ckpt-7.index"]
    r = tb.read_bundle(prefix)
    assert set(r) == set(v)
    for k in v:
        a = np.asarray(v[k])
        assert r[k].dtype == a.dtype and r[k].shape == a.shape and np.array_equal(r[k], a), k
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(prefix)
    d = str(tmp_path / "checkpoints")
    for step in range(7):
        ckpt.save(d, step, {"vae/mean/kernel": np.full((2, 3), step, np.float32), "vae/step_idx": np.int32(step)}, fmt="tf" if step % 2 else "both")
    assert sorted(f for f in os.listdir(d) if f.endswith(".index")) == ["model.ckpt-%d.index" % i for i in range(2, 7)]      # max_to_keep = 5
    assert sorted(f for f in os.listdir(d) if f.endswith(".npz")) == ["model.ckpt-%d.npz" % i for i in (2, 4, 6)]
    os.remove(os.path.join(d, "model.ckpt-6.npz"))                                 # what is left is exactly what TensorFlow would have written
    sd = ckpt.load(ckpt.latest(d))
    assert sd["vae/mean/kernel"][0, 0] == 6 and int(sd["vae/step_idx"]) == 6 and sd["vae/step_idx"].shape == ()


def test_tensorboard_event_files_match_the_reference_framing(golden_dir, tmp_path):
    """mi355/summary.py writes tf.summary.FileWriter files: the reference's own event file (first records, verbatim) passes our reader with
    both CRCs of every record checked, its first scalars are the ones in ref_event_scalars.json, and a file written here starts with the
    same file_version record (identical bytes apart from the wall-clock field and the two checksums that cover it) and reads back."""
    import glob
    from mi355 import summary as sm
    ref_path = os.path.join(golden_dir, "ref_events_head.bin")
    ver, series = sm.read_events(ref_path, verify=True)
    scal = json.load(open(os.path.join(golden_dir, "ref_event_scalars.json")))["vae_rgb/train"]
    assert ver == "brain.Event:2" and set(series) == {"vae/kl_loss", "vae/reconstruction_loss", "vae/learning_rate"}
    for tag, pts in series.items():
        assert pts[0][0] == scal[tag]["first_step"] and pts[0][2] == pytest.approx(scal[tag]["first"], rel=1e-6)
    w = sm.SummaryWriter(str(tmp_path / "logs"))
    for i in range(4):
        w.add_scalar("vae/kl_loss", 44.5 - i, i)
        w.add_scalar("vae/reconstruction_loss", 25000.0 / (i + 1), i)
    w.add_text("hyperparameters", {"learning_rate": 1e-4, "z_dim": 64}, 0)
    w.close()
    (path,) = glob.glob(str(tmp_path / "logs" / "events.out.tfevents.*"))
    ver2, s2 = sm.read_events(path, verify=True)
    assert ver2 == "brain.Event:2" and [p[0] for p in s2["vae/kl_loss"]] == [0, 1, 2, 3] and [p[2] for p in s2["vae/kl_loss"]] == [44.5, 43.5, 42.5, 41.5]
    assert s2["vae/reconstruction_loss"][3][2] == pytest.approx(6250.0)
    a, b = open(ref_path, "rb").read()[:40], open(path, "rb").read()[:40]
    assert a[:12] == b[:12] and a[12:13] == b[12:13] and a[21:36] == b[21:36]           # length, its crc, field tags, "brain.Event:2"
    raw = bytearray(open(path, "rb").read())
    raw[50] ^= 1
    open(path, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        sm.read_events(path)


def test_dataset_loader_is_parallel_ordered_and_uint8(tmp_path):
    """vae/train_vae.py load_images: the threaded decode returns exactly what the reference's serial loop returns -- every *.png in
    os.listdir order -- and the uint8 variant of the RGB preprocessing is the float one before its division (normalised later on the device)."""
    from PIL import Image
    import vae.train_vae as tv
    d = tmp_path / "rgb"
    d.mkdir()
    rng = np.random.RandomState(0)
    frames = {}
    for i in rng.permutation(23):
        a = rng.randint(0, 256, (80, 160, 4), dtype=np.uint8)
        Image.fromarray(a, "RGBA").save(str(d / ("%d.png" % i)))
        frames["%d.png" % i] = a
    (d / "notes.txt").write_text("not a frame")
    order = [f for f in os.listdir(str(d)) if f.endswith(".png")]
    serial = tv.load_images(str(d), tv.preprocess_rgb_frame, workers=1)
    par = tv.load_images(str(d), tv.preprocess_rgb_frame, workers=8)
    u8 = tv.load_images(str(d), tv.rgb_frame_u8, workers=8)
    assert serial.shape == (23, 80, 160, 3) and serial.dtype == np.float32 and np.array_equal(serial, par)
    assert u8.dtype == np.uint8 and np.array_equal(u8, np.stack([frames[f][:, :, :3] for f in order]))
    assert np.array_equal(u8.astype(np.float32) / 255.0, serial)
    tr, va = tv.train_val_split(u8, 0.1)
    assert len(va) == 2 and len(tr) == 21 and np.array_equal(va, u8[:2])
    with pytest.raises(FileNotFoundError):
        tv.load_images(str(tmp_path), tv.rgb_frame_u8)


def test_fastdiv_constants_are_exact():
    """Python mirror of make_fastdiv()/FastDiv::div (csrc/common.hpp): q = (n * mul) >> shift must equal n // d for n < 2^31."""
    rng = np.random.RandomState(0)
    for d in [1, 2, 3, 4, 5, 6, 7, 8, 12, 18, 19, 20, 32, 38, 39, 40, 48, 64, 79, 80, 144, 160, 256, 684, 720, 760, 800, 3081, 3200, 6144, 12800, 38400, 2 ** 20 + 7]:
        if d <= 1:
            mul, shift = 1, 0
        else:
            s = 0
            while (1 << s) < d:
                s += 1
            shift = 31 + s
            mul = ((1 << shift) // d) + 1
            assert mul < 2 ** 32
        ns = np.concatenate([np.arange(0, 4096), rng.randint(0, 2 ** 31 - 1, 20000), [2 ** 31 - 1, 2 ** 31 - 2], d * np.arange(1, 200) - 1, d * np.arange(1, 200)])
        ns = ns[(ns >= 0) & (ns < 2 ** 31)].astype(object)
        for n in ns:
            assert (int(n) * mul) >> shift == int(n) // d, (d, n)


def test_shard_bounds_partition_every_minibatch():
    from mi355 import dist as midist
    for n in (1, 7, 32, 100, 512, 4096):
        for w in (1, 2, 3, 8):
            spans = [midist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    idx = np.arange(10)
    assert np.array_equal(np.concatenate([midist.shard(idx, r, 4) for r in range(4)]), idx)


def test_data_parallel_two_ranks_gloo_equals_single_process(tmp_path):
    """world_size-2 gloo run of the product's DP scheme (mi355.dist) around the oracle's gradients: sharded rows +
    summed flat gradients + identical Adam == one process on the global minibatch."""
    script = os.path.join(ROOT, "tests", "dp_worker.py")
    out = str(tmp_path / "dp")
    os.makedirs(out)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", script, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(os.path.join(out, "result.json")))
    assert res["world"] == 2
    assert res["max_grad_rel_err"] < 2e-5, res
    assert res["max_param_diff_between_ranks"] == 0.0, res
    # Adam's first step is g/(|g|+eps): entries with |g| ~ 1e-8 amplify the 1e-7 summation-order difference
    assert res["max_param_rel_err_vs_single"] < 2e-2, res
    assert abs(res["metric_recon_dp"] / res["metric_recon_single"] - 1) < 1e-6
    assert res["ppo_max_grad_rel_err"] < 2e-5, res


def test_vae_common_call_chain_is_pinned_to_the_reference(golden_dir):
    """SURVEY 8 row a22.  tests/golden/vae_common_calls.json is what the reference's own vae_common.py (load_vae :6-27, encode_state :45-61) does
    to the `vae.models` classes -- recorded by running /root/reference/vae_common.py against a recording stand-in.  The restatement the GPU
    test drives the drop-in with (tests/ref_call_chain.py) must produce the identical trace; where the reference checkout exists, the
    fixture is re-derived from it."""
    import ref_call_chain as rc
    golden = json.load(open(os.path.join(golden_dir, "vae_common_calls.json")))
    assert sorted(golden) == sorted(rc.MODEL_DIRS)
    for d in rc.MODEL_DIRS:
        assert rc.trace_restatement(d) == golden[d], d
        if os.path.exists(os.path.join(rc.REFERENCE, "vae_common.py")):
            assert rc.trace_reference(d) == golden[d], d
    t = golden[rc.MODEL_DIRS[0]]
    assert [c["call"] for c in t["calls"]] == ["ConvVAE", "init_session", "load_latest_checkpoint", "encode"]
    assert t["calls"][0]["kwargs"]["training"] is False and t["calls"][0]["kwargs"]["models_dir"] == "vae"
    assert t["calls"][0]["kwargs"]["target_shape"]["ndarray"] == [80, 160, 1]                # "seg_" in the directory name
    assert t["state"] == {"shape": [67], "dtype": "float64", "head": [0.0, 1.0, 2.0], "tail": [-0.25, 0.5, 12.5]}
    # the drop-in's constructors accept exactly these keywords (models_dir is swallowed by **kwargs as in the reference, vae/models.py:41)
    from vae.models import ConvVAE, MlpVAE
    for cls in (ConvVAE, MlpVAE):
        sig = inspect.signature(cls.__init__)
        assert any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())


class _ScheduleDev:
    """Stand-in for VaeDevice in the collective-schedule test: the three gradient buckets of the real layout, no arithmetic."""
    def __init__(self):
        import torch
        self.device = torch.device("cpu")
        self.n_flat, self.P, self.source_shape = 2584392, 38400, (80, 160, 3)
        c4, dec = 165600, 1476704                       # conv4 kernel / dense1 kernel offsets of the flat layout (pad-8 TF creation order)
        self.grad_buckets = [(1, dec, self.n_flat), (3, c4, dec), (4, 0, c4)]
        self.grads, self.metrics, self.losses = torch.zeros(self.n_flat), torch.zeros(3), torch.zeros(2)
        self.log = []

    def set_seed(self, s): pass
    def forward(self, *a, **k): self.log.append(("fwd", a[3]))
    def backward(self, src, idx, eps, inv, part=0): self.log.append(("bwd", part))
    def apply_adam(self, *a): self.log.append(("adam",))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dp_collective_schedule_is_identical_on_every_rank(tmp_path, monkeypatch, world):
    """SURVEY 8e / VERDICT r02 #9: for every global batch B = 8 k every rank must issue the SAME sequence of collectives (op, element count) --
    three gradient-bucket all-reduces per SGD step in backward's completion order, one metric all-reduce per epoch -- whatever its rank, or the
    job hangs.  Also: PPO.train's logging path issues none (a rank that logs must not add a collective the others do not)."""
    import torch
    from mi355 import dist as midist
    from vae.models import ConvVAE
    per_rank = []
    for B in (8, 64, 512, 4096):
        seqs = []
        for r in range(world):
            calls = []
            monkeypatch.setattr(midist, "world_size", lambda: world)
            monkeypatch.setattr(midist, "rank", lambda r=r: r)

            class _W:
                def wait(self): calls.append(("wait",))

            def fake_allreduce(t, async_op=False):
                calls.append(("all_reduce_sum", int(t.numel()), bool(async_op)))
                return _W()
            monkeypatch.setattr(midist, "all_reduce_sum", fake_allreduce)
            m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / ("m%d_%d" % (B, r))), precision="bf16", seed=0)
            m.dev = _ScheduleDev()
            monkeypatch.setattr(m, "_frames", lambda arr, *a, **k: torch.zeros(len(arr), 4))
            np.random.seed(3)                                # every rank draws the same legacy-numpy permutation
            n = 2 * B + 3                                    # two full steps, remainder dropped
            table = np.zeros((n, 4), np.float32)
            m._epoch(table, table, B, True)
            fw = [x for x in m.dev.log if x[0] == "fwd"]
            assert [x[1] for x in fw] == [B // world] * 2    # equal shares of every global minibatch
            seqs.append(calls)
        assert all(s == seqs[0] for s in seqs[1:]), (B, world)
        per_step = [c for c in seqs[0] if c[0] == "all_reduce_sum" and c[2]]
        assert len(per_step) == 6 and sum(c[1] for c in per_step[:3]) == 2584392      # 3 buckets x 2 steps; the buckets tile the flat gradient buffer
        assert seqs[0][-1] == ("all_reduce_sum", 3, False)                            # the epoch's metric sums
        per_rank.append(seqs[0])
        # what each of those bucket all-reduces becomes INSIDE the library under either schedule (mi_comm_allreduce_plan: the pure function csrc/comm.hip's
        # allreduce_on executes): the same ops with the same element counts on every rank -- a rank that scattered while another all-reduced would hang.
        # rsag: W equal rank slices tiling the bucket's first chunk * W floats (slice r at offset r * chunk), the n % W tail as one small all-reduce.
        from mi355 import lib as milib
        L = milib.get()
        for (_, n, _async) in per_step[:3]:
            for algo in (0, 1):
                plans = []
                for r in range(world):
                    out = np.zeros(5, np.int64)
                    assert L.mi_comm_allreduce_plan(algo, world, r, n, out.ctypes.data) == 0
                    plans.append(out.copy())
                assert all(pl[0] == plans[0][0] and pl[1] == plans[0][1] and pl[3] == plans[0][3] and pl[4] == plans[0][4] for pl in plans), (n, algo)
                rsag, chunk, _, tail_off, tail_n = (int(x) for x in plans[0])
                if algo == 0 or n // world < 1024:
                    assert rsag == 0                       # one ncclAllReduce of n floats everywhere
                else:
                    assert rsag == 1 and chunk == n // world and [int(pl[2]) for pl in plans] == [r * chunk for r in range(world)]
                    assert tail_off == chunk * world and tail_n == n - chunk * world and 0 <= tail_n < world
    assert all(len(s) == len(per_rank[0]) for s in per_rank)
    # Round 5: the data-parallel step is ONE C call (mi_vae_train_step_dp) that walks the library's own bucket table (mi_vae_dp_buckets) and queues every bucket through
    # mi_allreduce_sum_f32_async on the communicator.  Replayed here on a RECORDING communicator (mi_comm_init_recording: no RCCL, no GPU) per rank and schedule: the
    # (op, floats, async) sequence the library issues is the one the host loop above issues, identical on every rank, and ends in one join of the three buckets.
    import ctypes
    from mi355 import lib as milib
    L = milib.get()
    d = milib.MiVaeDesc(1, 512, 80, 160, 3, 3, 64, 0, 1.0, 0.0)
    bk = np.zeros(9, np.int64)
    L.mi_vae_dp_buckets(ctypes.byref(d), bk.ctypes.data)
    buckets = [tuple(int(x) for x in bk[3 * i:3 * i + 3]) for i in range(3)]
    assert buckets == _ScheduleDev().grad_buckets
    host_step = [c for c in per_rank[0] if c[0] == "all_reduce_sum" and c[2]][:3]
    assert [hi - lo for (_, lo, hi) in buckets] == [c[1] for c in host_step]
    base = 1 << 20                                        # a made-up gradient-buffer address: a recording communicator never dereferences it
    for algo in (0, 1):
        logs = []
        for r in range(world):
            log = np.zeros((32, 4), np.int64)
            h = ctypes.c_void_p()
            L.mi_comm_init_recording(ctypes.addressof(h), r, world, log.ctypes.data, 32)
            L.mi_comm_set_algo(h, algo)
            for (_, lo, hi) in buckets:
                L.mi_allreduce_sum_f32_async(h, None, base + 4 * lo, hi - lo)
            L.mi_comm_wait(h, None)
            n = L.mi_comm_recorded(h)
            assert 0 < n <= 32
            L.mi_comm_destroy(h)
            logs.append(log[:n].copy())
        ops = [[(int(e[0]), int(e[1]), int(e[2])) for e in lg] for lg in logs]
        assert all(o == ops[0] for o in ops[1:]), (world, algo)                 # every rank: the same ops with the same element counts
        assert ops[0][-1] == (5, 3, 0)                                           # one join of the three buckets in front of the optimiser step
        if algo == 0:
            assert ops[0][:-1] == [(1, c[1], 1) for c in host_step]              # = the host loop's three async bucket all-reduces
        else:
            i = 0
            for (_, lo, hi) in buckets:
                n_b, chunk = hi - lo, (hi - lo) // world
                assert chunk >= 1024
                assert ops[0][i] == (2, chunk, 1) and ops[0][i + 1] == (3, chunk, 1)
                for r in range(world):                                           # rank r gathers ITS slice: offset r * chunk of the bucket
                    assert int(logs[r][i][3]) == base + 4 * lo and int(logs[r][i + 1][3]) == base + 4 * (lo + r * chunk)
                i += 2
                if n_b - chunk * world:
                    assert ops[0][i] == (1, n_b - chunk * world, 1)
                    i += 1
            assert i == len(ops[0]) - 1


def test_ppo_logging_path_issues_no_collective(tmp_path, monkeypatch):
    """PPO.train() on a rank with a summary writer must not issue a collective the non-logging ranks do not issue."""
    import torch
    import ppo as ppo_mod
    from mi355 import dist as midist
    src = inspect.getsource(ppo_mod.PPO.train)
    assert "_local_losses" in src and "_global_losses" not in src
    calls = []
    monkeypatch.setattr(midist, "world_size", lambda: 4)
    monkeypatch.setattr(midist, "all_reduce_sum", lambda *a, **k: calls.append(a))

    class Box:
        low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
    m = ppo_mod.PPO(np.array([67]), Box(), model_dir=str(tmp_path / "p"), seed=0)

    class Dev:
        losses = torch.tensor([0.25, 0.5, 0.01, 0.24, 0.25, 0.1, 0.2, 1.0, 1.0])
    m.dev = Dev()
    L = m._local_losses()
    assert not calls
    # sums over local rows / M_global -> x world = local means; entropy and std (state independent) untouched
    assert np.allclose(L[[0, 1, 4]], [1.0, 2.0, 1.0]) and np.allclose(L[5:7], [0.4, 0.8]) and np.allclose(L[7:9], [1.0, 1.0]) and np.isclose(L[2], 0.01)
    assert np.isclose(L[3], -1.0 + 2.0 - 0.01)


def test_gpu_free_entry_points_of_the_c_abi():
    """SURVEY 5 (sanitizer row): everything the C ABI does without a GPU -- descriptor / layout / workspace arithmetic incl. the guard mode, the error paths
    with missing, short or misaligned buffers, CRC32C over every length and alignment, tuning get / set -- driven by tools/asan_host_check.py.  Here against
    the product build; `tools/asan_host_check.sh` runs the same script against the AddressSanitizer build of the host side (build/asan/, minutes to build)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_host_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "asan host check: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_device_probe_argument_validation_without_a_gpu():
    """mi_device_probe (round 5, box calibration): refuses missing / short / misaligned buffers before it touches the device."""
    from mi355 import lib as milib
    L = milib.get()
    assert L.mi_device_probe_scratch_bytes() >= (512 << 20)
    out8 = np.zeros(8, np.float32)
    for scratch, nbytes, out in ((None, 1 << 30, out8.ctypes.data), (4096, 1 << 20, out8.ctypes.data), (4096 + 8, 1 << 30, out8.ctypes.data), (4096, 1 << 30, None)):
        with pytest.raises(milib.MiError):
            L.mi_device_probe(None, scratch, nbytes, 5, out)


def test_hand_scheduled_loads_of_the_fused_encoder_head_are_never_read_early():
    """csrc/enc12_tile.hpp (ring form, pipelined conv2): the frame loads and the LDS fragment reads of the production instantiations are inline assembly with hand-written
    s_waitcnt, so the compiler's own hazard tracking does not cover them.  tools/check_enc12_isa.py compiles enc12.hip to a gfx950 listing (no GPU needed) and replays
    both kernels under the hardware's rule -- loads retire in order, a wait leaves at most N outstanding -- twice around the band loop: no instruction reads a register
    whose load may still be in flight, no spills, the expected number of loads and waits.  (A register copy or a spill the compiler might one day place between a load and
    its wait would read stale data silently; this is the test that would notice.)"""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_enc12_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 violations") == 2, r.stdout
    # the same replay over the activation-resident kernels (ares_tile.hpp: LDS fragment reads and their waits by hand -- the idiom the fused encoder head took over)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_enc12_isa.py"), "--generic", os.path.join(root, "carla-ppo_amd", "csrc", "ares.hip"),
                        "ares_(conv|gather|gather2)_kernel"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 violations" in r.stdout, r.stdout + r.stderr


def test_bench_clock_conditioning_leaves_every_rank_after_the_same_number_of_steps(tmp_path):
    """bench.py's clock conditioning is a WALL-CLOCK loop of SGD steps, and in a data-parallel run every step is a collective: ranks that each read their own clock leave it
    after different numbers of steps and the job hangs (the two-rank bench test on the GPU did, once in seven runs -- and the driver's --gpus 8 launch would).
    bench.condition_clocks lets the ranks agree after every chunk.  Two gloo ranks on the CPU, one of them 3 ms per step slower: both return, after the same number of steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29561",
                        os.path.join(root, "tests", "condition_worker.py"), str(tmp_path)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(os.path.join(str(tmp_path), "cond_rank%d.json" % k))) for k in (0, 1))
    assert a["steps"] == b["steps"] and a["steps"] >= 5 and a["steps"] % 5 == 0, (a, b)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ppo_and_mlp_vae_dp_steps_are_one_c_call_with_the_same_collectives_on_every_rank(tmp_path, monkeypatch, world):
    """Round 6.  With the library's communicator live, a data-parallel PPO minibatch step is ONE call of the device (mi_ppo_train_step_dp) that keeps
    the in-kernel minibatch gather (before: forward_backward -> a blocking Python-side all-reduce -> apply_adam, rows gathered on the host whenever world_size > 1), and the
    MlpVAE step is one call as the ConvVAE's (before: the host loop).  No GPU here: stand-in devices record what the host mirror asks of them per rank; the collectives the C calls
    issue are replayed on RECORDING communicators (no RCCL) for every rank under both bucket schedules: same ops, same element counts everywhere."""
    import ctypes
    import torch
    from mi355 import dist as midist
    from mi355 import lib as milib
    import ppo as ppo_mod
    from oracle import ppo_oracle as po
    from vae.models import MlpVAE
    L = milib.get()

    class _Comm:
        handle = 0xC0FFEE
    monkeypatch.setattr(midist, "world_size", lambda: world)
    monkeypatch.setattr(midist, "mi_comm", lambda: _Comm)

    class _PpoDev:
        def __init__(self): self.calls = []; self.losses = torch.zeros(16); self.device = torch.device("cpu")
        def fused_ok(self): return True
        def train_step_dp(self, comm, s, a, r, adv, lp, rows, m, inv_m, gs, alpha, *rest): self.calls.append(("dp", comm, tuple(s.shape), None if rows is None else tuple(rows.shape), m, inv_m, gs))
        def train_step(self, *a, **k): self.calls.append(("single",))
        def train_step_idx(self, *a, **k): self.calls.append(("single_idx",))
        def forward_backward(self, *a, **k): self.calls.append(("fb",))
        def apply_adam(self, *a, **k): self.calls.append(("adam",))
    per_rank = []
    for r in range(world):
        monkeypatch.setattr(midist, "rank", lambda r=r: r)
        m = ppo_mod.PPO(np.array([67]), po.ActionSpace(), model_dir=str(tmp_path / ("p%d" % r)), seed=1)
        m.dev = _PpoDev()
        T, mb = 128, 32 // world if world <= 8 else 4
        s, a, R, A = torch.zeros(T, 67), torch.zeros(T, 2), torch.zeros(T), torch.zeros(T)
        rows = torch.arange(mb, dtype=torch.int32)
        b1 = m.beta1_power
        m._step_rows(s, a, R, A, None, rows, mb, mb * world)             # the replay's / train()'s resident form: rows of THIS rank's tables
        m._step_resident(s[:mb], a[:mb], R[:mb], A[:mb], mb, mb * world)   # PPO.train's host-minibatch form
        assert m.beta1_power == np.float32(np.float32(b1 * np.float32(0.9)) * np.float32(0.9))
        per_rank.append(m.dev.calls)
    want = [("dp", 0xC0FFEE, (128, 67), (32 // world,), 32 // world, 1.0 / 32, 1.0 / world), ("dp", 0xC0FFEE, (32 // world, 67), None, 32 // world, 1.0 / 32, 1.0 / world)]
    assert all(c == want for c in per_rank), per_rank[0]
    # MI355_DP_HOST_LOOP=1 and a process group without the library communicator (gloo) keep the host-sequenced form
    monkeypatch.setattr(midist, "mi_comm", lambda: None)
    monkeypatch.setattr(midist, "all_reduce_sum", lambda t, async_op=False: None)
    m = ppo_mod.PPO(np.array([67]), po.ActionSpace(), model_dir=str(tmp_path / "ph"), seed=1)
    m.dev = _PpoDev(); m.dev.grads = torch.zeros(4)
    m._step_resident(torch.zeros(4, 67), torch.zeros(4, 2), torch.zeros(4), torch.zeros(4), 4, 4 * world)
    assert m.dev.calls == [("fb",), ("adam",)]
    monkeypatch.setattr(midist, "mi_comm", lambda: _Comm)

    # the MlpVAE takes the one-call branch of VAE._train_minibatch as soon as its device offers train_step_dp
    class _MlpDev:
        def __init__(self): self.calls = []
        def train_step_dp(self, comm, src, tgt, idx, B, inv_batch, eps, alpha, *rest): self.calls.append(("dp", comm, B, inv_batch))
        def forward(self, *a, **k): self.calls.append(("fwd",))
    mv = MlpVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "mlp"), precision="bf16", seed=0)
    mv.dev = _MlpDev()
    mv._train_minibatch(None, None, None, 512 // world, 1.0 / 512, None)
    assert mv.dev.calls == [("dp", 0xC0FFEE, 512 // world, 1.0 / 512)]
    from mi355.mlp_vae_device import MlpVaeDevice
    from mi355.ppo_device import PpoDevice
    assert hasattr(MlpVaeDevice, "train_step_dp") and hasattr(PpoDevice, "train_step_dp")

    # what those C calls issue, per rank and bucket schedule: PPO = ONE in-stream all-reduce of the flat gradient buffer; MlpVAE = decoder bucket, encoder bucket (async), one join
    dp = milib.MiPpoDesc(256, 67, 2, 500, 300, 0.2, 1.0, 0.01)
    n_ppo = int(L.mi_ppo_param_floats(ctypes.byref(dp)))
    assert n_ppo >= 369505
    dm = milib.MiMlpVaeDesc()
    dm.dtype, dm.max_batch, dm.source_size, dm.target_size, dm.z_dim, dm.n_enc, dm.n_dec = 1, 512, 38400, 38400, 64, 2, 2
    dm.enc[0], dm.enc[1], dm.dec[0], dm.dec[1] = 512, 256, 256, 512
    dm.loss_kind, dm.with_optimizer, dm.beta, dm.kl_tolerance = 0, 1, 1.0, 0.0
    n_mlp = int(L.mi_mlpvae_param_floats(ctypes.byref(dm)))
    dec_off = n_mlp // 2                                  # (any split: the plan only depends on the bucket sizes)
    for sizes, asyncs in (([n_ppo], 0), ([n_mlp - dec_off, dec_off], 1)):
        for algo in (0, 1):
            logs = []
            for r in range(world):
                log = np.zeros((16, 4), np.int64)
                h = ctypes.c_void_p()
                L.mi_comm_init_recording(ctypes.addressof(h), r, world, log.ctypes.data, 16)
                L.mi_comm_set_algo(h, algo)
                for n in sizes:
                    (L.mi_allreduce_sum_f32_async if asyncs else L.mi_allreduce_sum_f32)(h, None, 1 << 20, n)
                L.mi_comm_wait(h, None)
                cnt = L.mi_comm_recorded(h)
                L.mi_comm_destroy(h)
                logs.append([(int(e[0]), int(e[1]), int(e[2])) for e in log[:cnt]])
            assert all(lg == logs[0] for lg in logs[1:]), (world, algo, sizes)
            if algo == 0:
                assert logs[0] == [(1, n, asyncs) for n in sizes] + ([(5, len(sizes), 0)] if asyncs else [])
            else:
                assert sum(e[1] * (world if e[0] == 2 else 1) for e in logs[0] if e[0] in (1, 2)) == sum(sizes)      # reduce-scatter slices x W + tails tile every bucket


def test_bench_line_helpers_step_traffic_and_watchdog():
    """bench.py round 6: `roofline.step_traffic` prices the committed PMC total of one whole step against SURVEY 8(d)'s two reference points, and the watchdog
    that guards the never-run-on-hardware collective schedule of an N > 1 bench fires its last words exactly once unless cancelled."""
    import importlib
    import threading
    bench = importlib.import_module("bench")
    st = bench.step_traffic(512)
    assert st["algorithmic_gb_survey_fp32_frames"] == pytest.approx(0.1408, rel=1e-3)          # 512 x (153,600 + 256) B + 2,584,387 x 24 B
    assert st["algorithmic_gb"] == pytest.approx(0.0818, rel=1e-3) and st["practical_unfused_gb"] == pytest.approx(1.598, rel=1e-3)
    assert st["pmc_gb_per_step"] and 1.5 < st["pmc_gb_per_step"] < 2.5 and st["source"].startswith("profiles/r06_pmc_traffic.json")
    assert st["x_practical"] == pytest.approx(st["pmc_gb_per_step"] / st["practical_unfused_gb"]) and st["x_algorithmic"] > 10
    # the watchdog: cancelled -> silent
    said = []
    t = bench._watchdog(0.2, lambda: said.append("line"))
    t.cancel()
    threading.Event().wait(0.4)
    assert said == []
    # ... and firing: last words, then os._exit (patched: the test process must survive)
    import os as _os
    real_exit, codes = _os._exit, []
    try:
        _os._exit = lambda c: codes.append(c)
        t = bench._watchdog(0.05, lambda: said.append("line"))
        t.join(2.0)
    finally:
        _os._exit = real_exit
    assert said == ["line"] and codes == [0]
