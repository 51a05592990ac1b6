"""MXNet NDArray-list (.params) reader / writer: round trips, legacy / V1 / V2 / V3 records built by oracle/mx_ndarray_format.py, error paths,
and the reference-named checkpoint helpers (lib/utils/load_model.py, save_model.py)."""
import struct

import numpy as np
import pytest

from lib.utils import mx_params as mxp
from lib.utils.load_model import load_checkpoint, load_param
from lib.utils.save_model import save_checkpoint


def test_round_trip_names_dtypes_shapes(tmp_path):
    rng = np.random.RandomState(0)
    data = {
        "arg:flow_conv1_weight": rng.randn(64, 8, 7, 7).astype(np.float32),
        "arg:flow_conv1_bias": rng.randn(64).astype(np.float32),
        "aux:bn_moving_mean": rng.randn(3, 1).astype(np.float64),
        "arg:idx": np.arange(12, dtype=np.int32).reshape(3, 4),
        "arg:u8": np.arange(7, dtype=np.uint8),
        "arg:i64": np.array([[2 ** 40, -5]], dtype=np.int64),
        "arg:half": rng.randn(2, 3).astype(np.float16),
    }
    f = str(tmp_path / "x-0001.params")
    mxp.nd_save(f, data)
    back = mxp.nd_load(f)
    assert list(back.keys()) == list(data.keys())  # order preserved like mx.nd.save
    for k in data:
        assert back[k].dtype == data[k].dtype and back[k].shape == data[k].shape
        np.testing.assert_array_equal(back[k], data[k])
    # header words
    raw = open(f, "rb").read()
    assert struct.unpack_from("<QQQ", raw, 0) == (0x112, 0, len(data))
    assert struct.unpack_from("<I", raw, 24)[0] == 0xF993FAC9


def test_list_without_names(tmp_path):
    f = str(tmp_path / "l.params")
    mxp.nd_save(f, [np.ones((2, 2), np.float32), np.zeros(3, np.float32)])
    back = mxp.nd_load(f)
    assert isinstance(back, list) and back[0].shape == (2, 2) and back[1].shape == (3,)


def _file(records, names):
    from oracle import mx_ndarray_format as fmt

    return fmt.file_bytes(records, names)


def test_writer_and_reader_against_the_oracle_format(tmp_path):
    """the package's writer against oracle/mx_ndarray_format.py's parser and byte builder (a second, independent statement of
    MXNet 1.2.0's NDArray::Save), and the package's reader against every generation the oracle can write"""
    from oracle import mx_ndarray_format as fmt

    rng = np.random.RandomState(3)
    data = {"arg:w": rng.randn(4, 3, 2, 2).astype(np.float32), "arg:b": rng.randn(4).astype(np.float32),
            "aux:m": rng.randn(2, 5).astype(np.float64), "arg:i": np.arange(6, dtype=np.int32).reshape(2, 3),
            "arg:h": rng.randn(3).astype(np.float16), "arg:u": np.arange(5, dtype=np.uint8), "arg:q": np.array([2 ** 40, -7], np.int64)}
    f = str(tmp_path / "w-0001.params")
    mxp.nd_save(f, data)
    raw = open(f, "rb").read()
    arrays, names = fmt.parse(raw)                                     # product writer -> oracle reader
    assert names == list(data)
    for a, k in zip(arrays, names):
        assert a.dtype == data[k].dtype and a.shape == data[k].shape
        np.testing.assert_array_equal(a, data[k])
    assert raw == fmt.file_bytes([fmt.record(v, "V2") for v in data.values()], list(data))   # and byte for byte what 1.2.0 writes
    for gen in ("legacy", "V1", "V2", "V3"):                            # oracle writer -> product reader
        for dev in ((1, 0), (2, 3)):                                    # saved from cpu(0) or gpu(3): the context is read and dropped
            g = tmp_path / "g.params"
            g.write_bytes(fmt.file_bytes([fmt.record(v, gen, *dev) for v in data.values()], list(data)))
            back = mxp.nd_load(str(g))
            assert list(back) == list(data)
            for k in data:
                assert back[k].dtype == data[k].dtype
                np.testing.assert_array_equal(back[k], data[k])
    g = tmp_path / "list.params"                                        # no names: mx.nd.load returns a list
    g.write_bytes(fmt.file_bytes([fmt.record(data["arg:w"], "V1")], []))
    back = mxp.nd_load(str(g))
    assert isinstance(back, list) and len(back) == 1
    np.testing.assert_array_equal(back[0], data["arg:w"])


def test_errors(tmp_path):
    f = tmp_path / "bad.params"
    f.write_bytes(struct.pack("<QQQ", 0x113, 0, 0))
    with pytest.raises(ValueError, match="invalid NDArray file"):
        mxp.nd_load(str(f))
    sparse = struct.pack("<Ii", 0xF993FAC9, 1)
    f.write_bytes(_file([sparse], ["arg:s"]))
    with pytest.raises(ValueError, match="sparse"):
        mxp.nd_load(str(f))
    a = np.ones(4, np.float32)
    trunc = struct.pack("<IiI1q", 0xF993FAC9, 0, 1, 4) + struct.pack("<ii", 1, 0) + struct.pack("<i", 0) + a.tobytes()[:8]
    f.write_bytes(_file([trunc], ["arg:t"]))
    with pytest.raises(ValueError, match="truncated"):
        mxp.nd_load(str(f))


def test_checkpoint_helpers_and_rename_rules(tmp_path):
    prefix = str(tmp_path / "deepim_ape")
    arg = {"fc6_weight": np.ones((2, 3), np.float32), "rot_weight_test": np.full((4, 2), 2.0, np.float32),
           "conv_i2r_weight": np.zeros(1, np.float32)}
    aux = {"stat": np.arange(3, dtype=np.float32)}
    name = save_checkpoint(prefix, 8, arg, aux)
    assert name.endswith("deepim_ape-0008.params")
    a2, x2 = load_checkpoint(prefix, 8)
    assert set(a2) == set(arg) and set(x2) == {"stat"}
    a3, _ = load_param(prefix, 8, process=True)
    assert "rot_weight" in a3 and "rot_weight_test" not in a3 and "conv_weight" in a3 and "conv_i2r_weight" not in a3
    np.testing.assert_array_equal(a3["rot_weight"], arg["rot_weight_test"])


def test_full_network_checkpoint_round_trip(tmp_path):
    """every parameter of the shipped LINEMOD graph survives save -> load bit for bit (57.7 M floats)"""
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from scene import make_test_config

    cfg = make_test_config()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=1)
    prefix = str(tmp_path / "full")
    save_checkpoint(prefix, 1, params, {})
    back, aux = load_checkpoint(prefix, 1)
    assert aux == {} and set(back) == set(params)
    for k in params:
        np.testing.assert_array_equal(back[k], params[k])


def test_init_weights_surgery_on_a_flownet_checkpoint():
    """deepIM_flownet.init_weights (:1009-1093): 6-channel first layer zero-extended to 8, heads re-initialised with
    init_from_flownet, everything else kept."""
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from scene import make_test_config

    cfg = make_test_config()
    cfg.network.init_from_flownet = True
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    rng = np.random.RandomState(3)
    ckpt = {"flow_conv1_weight": rng.randn(64, 6, 7, 7).astype(np.float32), "conv2_weight": rng.randn(128, 64, 5, 5).astype(np.float32),
            "fc6_weight": np.full((256, 81920), 7.0, np.float32), "trans_weight": np.ones((3, 256), np.float32)}
    keep = {k: v.copy() for k, v in ckpt.items()}
    params = sym.init_weights(cfg, ckpt, {}, seed=0)
    assert params["flow_conv1_weight"].shape == (64, 8, 7, 7)
    np.testing.assert_array_equal(params["flow_conv1_weight"][:, :6], keep["flow_conv1_weight"])
    assert np.all(params["flow_conv1_weight"][:, 6:] == 0)
    np.testing.assert_array_equal(params["conv2_weight"], keep["conv2_weight"])
    assert np.abs(params["fc6_weight"]).max() < 1.0          # xavier again, not the file's 7.0
    assert np.all(params["trans_weight"] == 0)               # :1074-1076
    assert np.all(params["rot_weight"][0] >= 0.01) and params["rot_weight"][1:].max() <= 0.01
    cfg.network.init_from_flownet = False
