"""SURVEY 8(f) row N4, on-disk I/O: the package's HDF5 reader / writer (`h5lite`), `MRISliceDataset` (reference data/mri_data.py:50-318)
and `save_reconstructions` (reference common/parts/utils.py:275-290).  CPU only.

The reader is pinned against files written by libhdf5 itself (tests/golden/h5_fastmri.h5 in the earliest format h5py defaults to,
h5_latest.h5 with libver = latest; tests/golden/make_h5_fixtures.py drives the C library through ctypes and regenerates the expected
arrays from its seed); the writer is pinned by reading its files back and -- where the image has it -- with libhdf5's own `h5dump`."""
import os
import random
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_h5_fixtures as fixtures  # noqa: E402

from mridc_amd.collections.common.parts import h5lite  # noqa: E402
from mridc_amd.collections.common.parts.utils import save_reconstructions  # noqa: E402
from mridc_amd.collections.reconstruction.data.mri_data import MRISliceDataset, et_query  # noqa: E402

H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


@pytest.mark.parametrize("fname", ["h5_fastmri.h5", "h5_latest.h5"])
def test_reader_vs_libhdf5_written_files(fname):
    want = fixtures.arrays()
    with h5lite.File(os.path.join(HERE, "golden", fname), "r") as f:
        assert sorted(f.keys()) == ["eta", "grp", "ismrmrd_header", "kspace", "mask", "reconstruction_rss", "sensitivity_map"]
        for k, v in want.items():
            ds = f[k]
            assert ds.shape == v.shape and ds.ndim == v.ndim and len(ds) == v.shape[0]
            got = ds[()]
            if k == "mask":                      # h5py's boolean enum over int8
                assert got.dtype == np.int8 and np.array_equal(got.astype(bool), v)
            else:
                assert got.dtype == v.dtype and np.array_equal(got, v), k
            assert np.array_equal(ds[1], np.asarray(v[1], dtype=got.dtype)) and np.array_equal(np.asarray(ds)[..., ::2], got[..., ::2])
        assert f["ismrmrd_header"][()] == fixtures.XML and f["ismrmrd_header"].shape == ()
        attrs = dict(f.attrs)
        assert attrs["max"] == 0.00123 and attrs["norm"] == 0.0456 and attrs["patient_id"] == "0beefc0ffeeé" and attrs["acquisition"] == "AXT2"
        assert "kspace" in f and "grp/counts" in f and "nope" not in f and "grp/nope" not in f
        assert f["grp"].keys() == ["counts"]
        with pytest.raises(KeyError):
            f["missing"]


def test_reader_rejects_what_it_does_not_parse(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(OSError):
        h5lite.File(p, "r")
    with pytest.raises(ValueError):
        h5lite.File(p, "a")


def _volume(rng, S=4, C=3, H=10, W=8):
    k = (rng.standard_normal((S, C, H, W)) + 1j * rng.standard_normal((S, C, H, W))).astype(np.complex64)
    return k


def test_writer_round_trip_and_h5dump(tmp_path):
    rng = np.random.default_rng(1)
    k = _volume(rng)
    arrays = {"kspace": k, "reconstruction": np.abs(k[:, 0]).astype(np.float32), "mask": rng.random((10, 8)) < 0.3,
              "f64": rng.standard_normal(7), "i64": np.arange(6).reshape(2, 3), "c128": k[0, 0].astype(np.complex128), "u8": np.arange(5, dtype=np.uint8),
              "scalar": np.float32(2.5)}
    p = tmp_path / "w.h5"
    with h5lite.File(p, "w") as f:
        for n, a in arrays.items():
            f.create_dataset(n, data=a)
        f.attrs["max"] = np.float64(3.5)
        f.attrs["patient_id"] = "abc123"
        f.attrs["vec"] = np.arange(4, dtype=np.int32)
        assert "kspace" in f and sorted(f.keys()) == sorted(arrays)
        with pytest.raises(ValueError):
            f.create_dataset("kspace", data=k)
    with h5lite.File(p, "r") as f:
        assert sorted(f.keys()) == sorted(arrays)
        for n, a in arrays.items():
            got = f[n][()]
            a = np.asarray(a)
            assert np.array_equal(got, a.astype(np.int8) if a.dtype == bool else a), n
            assert f[n].shape == a.shape
        at = f.attrs
        assert at["max"] == 3.5 and at["patient_id"] == "abc123" and np.array_equal(at["vec"], np.arange(4))
    if H5DUMP is None:
        pytest.skip("no h5dump in this image: the libhdf5 cross-check of the writer is skipped")
    hdr = subprocess.run([H5DUMP, "-H", str(p)], capture_output=True, text=True)
    assert hdr.returncode == 0 and 'H5T_IEEE_F32LE "r"' in hdr.stdout and "( 4, 3, 10, 8 )" in hdr.stdout, hdr.stderr
    out = subprocess.run([H5DUMP, "-d", "/f64", "-m", "%.17g", str(p)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    nums = [float(t.strip(",")) for t in out.stdout.split("DATA {")[1].split("}")[0].split() if not t.startswith("(")]
    assert np.array_equal(np.array(nums), arrays["f64"])
    at = subprocess.run([H5DUMP, "-a", "/patient_id", str(p)], capture_output=True, text=True)
    assert at.returncode == 0 and '"abc123"' in at.stdout


HEADER = ('<?xml version="1.0"?><ismrmrdHeader xmlns="{ns}"><encoding><encodedSpace><matrixSize><x>20</x><y>11</y><z>1</z></matrixSize>'
          "</encodedSpace><reconSpace><matrixSize><x>10</x><y>8</y><z>1</z></matrixSize></reconSpace><encodingLimits><kspace_encoding_step_1>"
          "<minimum>0</minimum><maximum>7</maximum><center>4</center></kspace_encoding_step_1></encodingLimits></encoding></ismrmrdHeader>")


def _make_dir(tmp_path, rng, ns="https://www.ismrm.org/ISMRMRD"):
    root = tmp_path / "multicoil_val"
    root.mkdir()
    vols = {}
    for i, S in enumerate((4, 3)):
        k = _volume(rng, S=S)
        sm = _volume(rng, S=S)
        tgt = rng.standard_normal((S, 10, 8)).astype(np.float32)
        mask = (rng.random((10, 8)) < 0.4).astype(np.float32)
        with h5lite.File(root / f"file{i}.h5", "w") as f:
            f.create_dataset("kspace", data=k)
            f.create_dataset("sensitivity_map", data=np.transpose(sm, (0, 2, 3, 1)))      # stored [S,H,W,C]: transposed back by the dataset
            f.create_dataset("mask", data=mask)
            f.create_dataset("reconstruction_sense", data=tgt)
            f.create_dataset("ismrmrd_header", data=HEADER.format(ns=ns))
            f.attrs["max"] = float(np.abs(tgt).max())
            f.attrs["acquisition"] = "AXT2"
        vols[f"file{i}.h5"] = (k, sm, mask, tgt)
    return root, vols


def test_mri_slice_dataset(tmp_path):
    rng = np.random.default_rng(3)
    root, vols = _make_dir(tmp_path, rng)
    ds = MRISliceDataset(root, challenge="multicoil")
    assert len(ds) == 7
    seen = []
    for i in range(len(ds)):
        kspace, smap, mask, eta, target, attrs, fname, sl = ds[i]
        k, sm, m, tgt = vols[fname]
        seen.append((fname, sl))
        assert kspace.dtype == np.complex64 and np.array_equal(kspace, k[sl])
        assert smap.shape == kspace.shape and np.array_equal(smap, sm[sl])
        assert np.array_equal(mask, m) and eta.size == 0 and np.array_equal(target, tgt[sl])
        assert attrs["acquisition"] == "AXT2" and attrs["encoding_size"] == (20, 11, 1) and attrs["recon_size"] == (10, 8, 1)
        assert attrs["padding_left"] == 11 // 2 - 4 and attrs["padding_right"] == 11 // 2 - 4 + 8
    assert seen == [("file0.h5", s) for s in range(4)] + [("file1.h5", s) for s in range(3)]
    # a transform receives the 8-tuple
    ds_t = MRISliceDataset(root, challenge="multicoil", transform=lambda *a: len(a))
    assert ds_t[0] == 8
    # consecutive slices: windows of 2, one fewer example per volume
    ds2 = MRISliceDataset(root, challenge="multicoil", consecutive_slices=2)
    assert len(ds2) == 5
    kspace, smap, *_ = ds2[1]
    assert np.array_equal(kspace, vols["file0.h5"][0][1:3]) and np.array_equal(smap, vols["file0.h5"][1][1:3])
    # sampling is driven by the `random` module, as in the reference
    random.seed(5)
    a = [(e[0].name, e[1]) for e in MRISliceDataset(root, challenge="multicoil", sample_rate=0.5).examples]
    random.seed(5)
    ex = [(f, s) for f in ("file0.h5", "file1.h5") for s in range(4 if f == "file0.h5" else 3)]
    random.shuffle(ex)
    assert a == ex[:round(7 * 0.5)]
    random.seed(6)
    assert {e[0].name for e in MRISliceDataset(root, challenge="multicoil", volume_sample_rate=0.5).examples} in ({"file0.h5"}, {"file1.h5"})
    assert len(MRISliceDataset(root, challenge="multicoil", num_cols=(11,))) == 7 and len(MRISliceDataset(root, challenge="multicoil", num_cols=(12,))) == 0
    # the metadata cache round-trips through plain YAML
    cache = tmp_path / "cache.yaml"
    first = MRISliceDataset(root, challenge="multicoil", use_dataset_cache=True, dataset_cache_file=cache)
    again = MRISliceDataset(root, challenge="multicoil", use_dataset_cache=True, dataset_cache_file=cache)
    assert cache.exists() and [(str(a_), b_, c_) for a_, b_, c_ in first.examples] == [(str(a_), b_, c_) for a_, b_, c_ in again.examples]
    with pytest.raises(ValueError):
        MRISliceDataset(root, challenge="knee")
    with pytest.raises(ValueError):
        MRISliceDataset(root, sample_rate=0.5, volume_sample_rate=0.5)
    with pytest.raises(ValueError):
        MRISliceDataset(root, consecutive_slices=0)


def test_header_namespace_quirk_and_external_maps(tmp_path):
    """The reference queries the https spelling of the ISMRMRD namespace (mri_data.py:21): a header declaring http:// answers "0"
    everywhere.  Sensitivity maps / masks may live in separate trees (sense_root / mask_root)."""
    from xml.etree.ElementTree import fromstring
    assert et_query(fromstring(HEADER.format(ns="http://www.ismrm.org/ISMRMRD")), ["encoding", "encodedSpace", "matrixSize", "y"]) == "0"
    assert et_query(fromstring(HEADER.format(ns="https://www.ismrm.org/ISMRMRD")), ["encoding", "encodedSpace", "matrixSize", "y"]) == "11"
    rng = np.random.default_rng(4)
    k = _volume(rng, S=2)
    sm = _volume(rng, S=2)
    m = (rng.random((10, 8)) < 0.5).astype(np.float32)
    root = tmp_path / "data" / "val"
    root.mkdir(parents=True)
    with h5lite.File(root / "v.h5", "w") as f:
        f.create_dataset("kspace", data=k)
        f.create_dataset("ismrmrd_header", data=HEADER.format(ns="http://www.ismrm.org/ISMRMRD"))
    (tmp_path / "sense" / "val").mkdir(parents=True)
    with h5lite.File(tmp_path / "sense" / "val" / "v.h5", "w") as f:
        f.create_dataset("sense", data=sm)
    (tmp_path / "masks").mkdir()
    with h5lite.File(tmp_path / "masks" / "v.h5", "w") as f:
        f.create_dataset("mask", data=m)
    ds = MRISliceDataset(root, challenge="multicoil", sense_root=tmp_path / "sense", mask_root=tmp_path / "masks")
    kspace, smap, mask, eta, target, attrs, fname, sl = ds[1]
    assert np.array_equal(kspace, k[1]) and np.array_equal(smap, sm[1]) and np.array_equal(mask, m) and target is None
    assert attrs["encoding_size"] == (0, 0, 0) and attrs["padding_left"] == 0 and attrs["padding_right"] == 1
    bare = MRISliceDataset(root, challenge="multicoil")
    with pytest.raises(ValueError):              # no maps anywhere: an empty array cannot be matched to k-space (mri_data.py:289-297)
        bare[0]


def test_save_reconstructions(tmp_path):
    rng = np.random.default_rng(9)
    recons = {"a.h5": rng.standard_normal((3, 10, 8)).astype(np.float32), "b.h5": (rng.standard_normal((2, 6, 6)) + 1j).astype(np.complex64)}
    save_reconstructions(recons, tmp_path / "out" / "reconstructions")
    for n, v in recons.items():
        with h5lite.File(tmp_path / "out" / "reconstructions" / n, "r") as f:
            assert f.keys() == ["reconstruction"] and np.array_equal(f["reconstruction"][()], v)


def test_runner_writes_reconstructions_like_test_epoch_end(tmp_path):
    """models/base.py:575-587: outputs grouped per file, stacked in slice order, one `reconstruction` dataset per volume."""
    from mridc_amd.runner import ReconstructionRunner
    rng = np.random.default_rng(11)
    vol = (rng.standard_normal((3, 1, 6, 5)) + 1j * rng.standard_normal((3, 1, 6, 5))).astype(np.complex64)
    outputs = [("v.h5", 2, vol[2]), ("v.h5", 0, vol[0]), ("w.h5", 0, vol[1]), ("v.h5", 1, vol[1])]
    stacked = ReconstructionRunner.save_outputs(outputs, tmp_path)
    assert np.array_equal(stacked["v.h5"], vol)
    with h5lite.File(tmp_path / "reconstructions" / "v.h5", "r") as f:
        assert np.array_equal(f["reconstruction"][()], vol)
    with h5lite.File(tmp_path / "reconstructions" / "w.h5", "r") as f:
        assert f["reconstruction"].shape == (1, 1, 6, 5)
