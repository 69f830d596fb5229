"""Drop-in for `mridc.collections.reconstruction.data.mri_data` (reference data/mri_data.py:21-318): `et_query` and `MRISliceDataset`,
the on-disk side of the reconstruction path.  HDF5 access goes through `h5lite` (this package's reader; h5py is not a dependency), the
ISMRMRD header through the standard library's ElementTree.  `__getitem__` returns the reference's 8-tuple
(kspace, sensitivity_map, mask, eta, target, attrs, fname, slice) or hands it to `transform` -- e.g. this package's
`MRIDataTransforms` (parts/transforms.py), which moves it to the GPU and yields the 9-tuple `ReconstructionRunner.test_step` takes."""
import logging
import os
import random
from pathlib import Path
from typing import Callable, Optional, Sequence, Tuple, Union
from xml.etree.ElementTree import fromstring

import numpy as np
import yaml
from torch.utils.data import Dataset

import mridc_amd.collections.common.parts.utils as utils
from mridc_amd.collections.common.parts import h5lite


def et_query(root, qlist: Sequence[str], namespace: str = "https://www.ismrm.org/ISMRMRD") -> str:
    """mri_data.py:21-47: text of the element at the end of the path `qlist`, "0" when it is absent.  (The default namespace is spelt
    with https as in the reference; headers declaring http://www.ismrm.org/ISMRMRD therefore answer "0" there too.)"""
    s = "."
    prefix = "ismrmrd_namespace"
    ns = {prefix: namespace}
    for el in qlist:
        s += f"//{prefix}:{el}"
    value = root.find(s, ns)
    if value is None:
        return "0"
    return str(value.text)


class MRISliceDataset(Dataset):
    """A dataset that loads slices from the .h5 volumes of one directory (mri_data.py:50-318)."""

    def __init__(self, root: Union[str, Path, os.PathLike], challenge: str = "segmentation", transform: Optional[Callable] = None,
                 sense_root: Union[str, Path, os.PathLike] = None, use_dataset_cache: bool = False, sample_rate: Optional[float] = None,
                 volume_sample_rate: Optional[float] = None, dataset_cache_file: Union[str, Path, os.PathLike] = "dataset_cache.yaml",
                 num_cols: Optional[Tuple[int]] = None, mask_root: Union[str, Path, os.PathLike] = None, consecutive_slices: int = 1):
        if challenge not in ("singlecoil", "multicoil", "segmentation"):
            raise ValueError('challenge should be either "singlecoil" or "multicoil" or "segmentation"')
        self.challenge = challenge
        if sample_rate is not None and volume_sample_rate is not None:
            raise ValueError("either set sample_rate (sample by slices) or volume_sample_rate (sample by volumes) but not both")
        self.sense_root = sense_root
        self.mask_root = mask_root
        self.dataset_cache_file = Path(dataset_cache_file)
        self.transform = transform
        self.recons_key = "reconstruction_esc" if challenge == "singlecoil" else "reconstruction_rss"
        self.examples = []
        if sample_rate is None:
            sample_rate = 1.0
        if volume_sample_rate is None:
            volume_sample_rate = 1.0

        # the cache holds plain types (file name, slice index, metadata) so that it loads with yaml.safe_load
        if self.dataset_cache_file.exists() and use_dataset_cache:
            with open(self.dataset_cache_file, "rb") as f:
                dataset_cache = yaml.safe_load(f) or {}
        else:
            dataset_cache = {}
        key = str(root)
        if dataset_cache.get(key) is None or not use_dataset_cache:
            files = list(Path(root).iterdir())
            for fname in sorted(files):
                metadata, num_slices = self._retrieve_metadata(fname)
                if not utils.is_none(num_slices) and not utils.is_none(consecutive_slices):
                    num_slices = num_slices - (consecutive_slices - 1)
                self.examples += [(fname, slice_ind, metadata) for slice_ind in range(num_slices)]
            if dataset_cache.get(key) is None and use_dataset_cache:
                dataset_cache[key] = [[str(f), int(s), {k: (list(v) if isinstance(v, tuple) else v) for k, v in m.items()}]
                                      for f, s, m in self.examples]
                logging.info(f"Saving dataset cache to {self.dataset_cache_file}.")
                with open(self.dataset_cache_file, "w") as f:
                    yaml.safe_dump(dataset_cache, f)
        else:
            logging.info(f"Using dataset cache from {self.dataset_cache_file}.")
            self.examples = [(Path(f), s, {k: (tuple(v) if isinstance(v, list) else v) for k, v in m.items()})
                             for f, s, m in dataset_cache[key]]

        if sample_rate < 1.0:                        # sample by slice
            random.shuffle(self.examples)
            self.examples = self.examples[:round(len(self.examples) * sample_rate)]
        elif volume_sample_rate < 1.0:               # sample by volume
            vol_names = sorted(list({f[0].stem for f in self.examples}))
            random.shuffle(vol_names)
            sampled_vols = vol_names[:round(len(vol_names) * volume_sample_rate)]
            self.examples = [example for example in self.examples if example[0].stem in sampled_vols]
        if num_cols:
            self.examples = [ex for ex in self.examples if ex[2]["encoding_size"][1] in num_cols]
        self.consecutive_slices = consecutive_slices
        if self.consecutive_slices < 1:
            raise ValueError("consecutive_slices value is out of range, must be > 0.")

    @staticmethod
    def _retrieve_metadata(fname):
        """mri_data.py:163-212: encoding / reconstruction matrix sizes and the k-space padding from the ISMRMRD header, the slice count
        from `kspace` (or `reconstruction`)."""
        with h5lite.File(fname, "r") as hf:
            if "ismrmrd_header" in hf:
                et_root = fromstring(hf["ismrmrd_header"][()])
                enc = ["encoding", "encodedSpace", "matrixSize"]
                enc_size = (int(et_query(et_root, enc + ["x"])), int(et_query(et_root, enc + ["y"])), int(et_query(et_root, enc + ["z"])))
                rec = ["encoding", "reconSpace", "matrixSize"]
                recon_size = (int(et_query(et_root, rec + ["x"])), int(et_query(et_root, rec + ["y"])), int(et_query(et_root, rec + ["z"])))
                params = ["encoding", "encodingLimits", "kspace_encoding_step_1"]
                enc_limits_center = int(et_query(et_root, params + ["center"]))
                enc_limits_max = int(et_query(et_root, params + ["maximum"])) + 1
                padding_left = int(enc_size[1] / 2) - enc_limits_center          # trunc division, as torch.div(..., rounding_mode="trunc")
                padding_right = padding_left + enc_limits_max
            else:
                padding_left = 0
                padding_right = 0
                enc_size = 0
                recon_size = (0, 0)
            num_slices = hf["kspace"].shape[0] if "kspace" in hf else hf["reconstruction"].shape[0]
        metadata = {"padding_left": padding_left, "padding_right": padding_right, "encoding_size": enc_size, "recon_size": recon_size}
        return metadata, num_slices

    def get_consecutive_slices(self, data, key, dataslice):
        """mri_data.py:214-243."""
        data = data[key]
        if self.consecutive_slices == 1:
            if data.shape[0] == 1:
                return data[0]
            if data.ndim != 2:
                return data[dataslice]
            return data
        num_slices = data.shape[0]
        if self.consecutive_slices > num_slices:
            return np.stack(data, axis=0)
        start_slice = dataslice
        end_slice = dataslice + self.consecutive_slices if dataslice + self.consecutive_slices <= num_slices else num_slices
        return data[start_slice:end_slice]

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i: int):
        """mri_data.py:248-318."""
        fname, dataslice, metadata = self.examples[i]
        with h5lite.File(fname, "r") as hf:
            kspace = self.get_consecutive_slices(hf, "kspace", dataslice).astype(np.complex64)
            if "sensitivity_map" in hf:
                sensitivity_map = self.get_consecutive_slices(hf, "sensitivity_map", dataslice).astype(np.complex64)
            elif self.sense_root is not None and self.sense_root != "None":
                with h5lite.File(Path(self.sense_root) / Path(str(fname).split("/")[-2]) / fname.name, "r") as sf:
                    if "sensitivity_map" in sf or "sensitivity_map" in next(iter(sf.keys())):
                        sensitivity_map = self.get_consecutive_slices(sf, "sensitivity_map", dataslice)
                    else:
                        sensitivity_map = self.get_consecutive_slices(sf, "sense", dataslice)
                    sensitivity_map = np.asarray(sensitivity_map).squeeze().astype(np.complex64)
            else:
                sensitivity_map = np.array([])
            if "mask" in hf:
                mask = np.asarray(self.get_consecutive_slices(hf, "mask", dataslice))
                if mask.ndim == 3:
                    mask = mask[dataslice]
            elif self.mask_root is not None and self.mask_root != "None":
                with h5lite.File(Path(self.mask_root) / fname.name, "r") as mf:
                    mask = np.asarray(self.get_consecutive_slices(mf, "mask", dataslice))
            else:
                mask = None
            eta = self.get_consecutive_slices(hf, "eta", dataslice).astype(np.complex64) if "eta" in hf else np.array([])
            if "reconstruction_sense" in hf:
                self.recons_key = "reconstruction_sense"
            target = self.get_consecutive_slices(hf, self.recons_key, dataslice) if self.recons_key in hf else None
            target = np.asarray(target) if target is not None else None
            attrs = dict(hf.attrs)
            attrs.update(metadata)

        if sensitivity_map.shape != kspace.shape:
            if sensitivity_map.ndim == 3:
                sensitivity_map = np.transpose(sensitivity_map, (2, 0, 1))
            elif sensitivity_map.ndim == 4:
                sensitivity_map = np.transpose(sensitivity_map, (0, 3, 1, 2))
            else:
                raise ValueError(f"Sensitivity map has invalid dimensions {sensitivity_map.shape} compared to kspace {kspace.shape}")

        item = (kspace, sensitivity_map, mask, eta, target, attrs, fname.name, dataslice)
        return item if self.transform is None else self.transform(*item)
