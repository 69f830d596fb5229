"""Drop-in for `mridc.collections.reconstruction.data.subsample` (reference subsample.py:13-664): k-space sub-sampling masks.

Host-side index work, once per slice (the reference runs it in its CPU data workers); the masks feed `apply_mask`
(`mrx_apply_mask`) and the cascades on the GPU.  The point of this module is **bit-identical masks**: every generator draws from
NumPy's legacy `RandomState` in exactly the order the reference does, so the same seed (the reference seeds with
`tuple(map(ord, fname))`, parts/transforms.py:290) gives the same mask bit for bit (tests/golden g12).

* `RandomMaskFunc`, `Equispaced1DMaskFunc`, `Equispaced2DMaskFunc` — seeded through the object's own `RandomState` (reproducible).
* `Gaussian1DMaskFunc`, `Gaussian2DMaskFunc` — the reference samples these with the *global* `np.random.choice`
  (subsample.py:353,438); the `seed` argument is ignored there too.  Same here: identical masks for identical global state.
* `Poisson2DMaskFunc` — the reference runs its dart throwing under `numba.jit(nopython=True)` (subsample.py:549-633), i.e. on
  Numba's private generator, which is not reproducible from any seed even there.  Here the same dart-throwing loop is a host routine of
  libmridc_amd on an explicit generator seeded through the object's `RandomState`: reproducible, equal to the reference's masks in law
  (acceleration, calibration block, centre disc, exclusion distances are tested), not bit for bit.
"""
import contextlib
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

__all__ = ["MaskFunc", "RandomMaskFunc", "Equispaced1DMaskFunc", "Equispaced2DMaskFunc", "Gaussian1DMaskFunc",
           "Gaussian2DMaskFunc", "Poisson2DMaskFunc", "create_mask_for_mask_type", "temp_seed"]

Seed = Optional[Union[int, Tuple[int, ...]]]


@contextlib.contextmanager
def temp_seed(rng, seed: Seed):
    """Seed `rng` for the duration of the block and put its previous state back afterwards (subsample.py:13-38)."""
    if seed is None:
        yield
        return
    saved = rng.get_state()
    rng.seed(seed)
    try:
        yield
    finally:
        rng.set_state(saved)


def _as_mask_tensor(values: np.ndarray, shape: Sequence[int], axes: Sequence[int]) -> torch.Tensor:
    """float32 tensor with singleton dims everywhere except `axes` (negative indices into `shape`)."""
    dims = [1] * len(shape)
    for ax, n in zip(axes, values.shape):
        dims[ax] = n
    return torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32).reshape(dims))


def _centre_block(n: int, fraction: float) -> Tuple[int, int]:
    """(start, length) of the fully sampled low-frequency block of an axis of n samples."""
    length = int(round(n * fraction))
    return (n - length + 1) // 2, length


class MaskFunc:
    """Base class (subsample.py:41-94): pairs of (center fraction, acceleration), one drawn uniformly per call."""

    def __init__(self, center_fractions: Sequence[float], accelerations: Sequence[int]):
        if len(center_fractions) != len(accelerations):
            raise ValueError("Number of center fractions should match number of accelerations")
        self.center_fractions = center_fractions
        self.accelerations = accelerations
        self.rng = np.random.RandomState()  # pylint: disable=no-member

    def __call__(self, shape: Sequence[int], seed: Seed = None, half_scan_percentage: Optional[float] = 0.0,
                 scale: Optional[float] = 0.02) -> Tuple[torch.Tensor, int]:
        raise NotImplementedError

    def choose_acceleration(self):
        pick = self.rng.randint(0, len(self.accelerations))
        return self.center_fractions[pick], self.accelerations[pick]


def _check_rank(shape):
    if len(shape) < 3:
        raise ValueError("Shape should have 3 or more dimensions")


class RandomMaskFunc(MaskFunc):
    """1-D random columns (subsample.py:96-155): the centre block plus every other column with probability
    (N/acc - N_low) / (N - N_low)."""

    def __call__(self, shape, seed: Seed = None, half_scan_percentage=0.0, scale=0.02):
        _check_rank(shape)
        with temp_seed(self.rng, seed):
            n = int(shape[-2])
            fraction, acceleration = self.choose_acceleration()
            start, n_low = _centre_block(n, fraction)
            keep_probability = (n / acceleration - n_low) / (n - n_low)
            columns = self.rng.uniform(size=n) < keep_probability
            columns[start:start + n_low] = True
        return _as_mask_tensor(columns, shape, [-2]), acceleration


class Equispaced1DMaskFunc(MaskFunc):
    """1-D (nearly) equispaced columns with a random offset (subsample.py:158-222); the spacing is adjusted for the centre block,
    so samples are rounded positions of a fractional stride, as in the public fastMRI masks."""

    def __call__(self, shape, seed: Seed = None, half_scan_percentage=0.0, scale=0.02):
        _check_rank(shape)
        with temp_seed(self.rng, seed):
            fraction, acceleration = self.choose_acceleration()
            n = int(shape[-2])
            start, n_low = _centre_block(n, fraction)
            columns = np.zeros(n, dtype=np.float32)
            columns[start:start + n_low] = 1
            stride = (acceleration * (n_low - n)) / (n_low * acceleration - n)
            first = self.rng.randint(0, round(stride))
            picks = np.around(np.arange(first, n - 1, stride)).astype(np.uint)
            columns[picks] = 1
        return _as_mask_tensor(columns, shape, [-2]), acceleration


class Equispaced2DMaskFunc(MaskFunc):
    """2-D lattice (subsample.py:225-281): per-axis acceleration = acc / 2, a centre rectangle of half the centre fraction per
    axis, lattice points at the truncated multiples of the per-axis acceleration.  Returns the full acceleration."""

    def __call__(self, shape, seed: Seed = None, half_scan_percentage=0.0, scale=0.02):
        _check_rank(shape)
        with temp_seed(self.rng, seed):
            fraction, acceleration = self.choose_acceleration()
            step, fraction = acceleration / 2, fraction / 2
            n_cols, n_rows = int(shape[-2]), int(shape[-3])
            c0, c_len = _centre_block(n_cols, fraction)
            r0, r_len = _centre_block(n_rows, fraction)
            grid = np.zeros((n_rows, n_cols), dtype=np.float32)
            grid[r0:r0 + r_len, c0:c0 + c_len] = 1
            rows = np.arange(0, n_rows, step).astype(int)
            cols = np.arange(0, n_cols, step).astype(int)
            grid[np.ix_(rows, cols)] = 1
        return _as_mask_tensor(grid, shape, [-3, -2]), step * 2


def _fwhm_pair(value):
    return value if isinstance(value, list) else [value] * 2


def _gaussian_profile(fwhm: float, n: int) -> np.ndarray:
    sigma = fwhm / np.sqrt(8 * np.log(2))
    x = np.linspace(-1.0, 1.0, n)
    return np.exp(-(x ** 2 / (2 * sigma ** 2)))


class Gaussian1DMaskFunc(MaskFunc):
    """Whole rows drawn without replacement from a Gaussian density over the first spatial axis, plus a fully sampled band of
    `scale` x the axis (subsample.py:284-373).  The center fractions act as FWHM.  Global `np.random` state, like the reference."""

    def __call__(self, shape, seed: Seed = None, half_scan_percentage=0.0, scale=0.02):
        self.shape = tuple(int(v) for v in shape[-3:-1])
        fwhm, acceleration = self.choose_acceleration()
        self.full_width_half_maximum = _fwhm_pair(fwhm)
        self.acceleration = acceleration
        self.scale = scale
        n0, n1 = self.shape
        band = int(n0 * scale)
        top = (n0 - band) // 2
        grid = np.zeros((n0, n1))
        grid[top:top + band, :] = 1.0
        density = _gaussian_profile(self.full_width_half_maximum[0], n0)
        density = density / density.sum()
        rows = np.random.choice(range(n0), size=int(n0 / acceleration), replace=False, p=density)
        grid[rows, :] = 1.0
        # the reference applies ifftshift along axis 0 twice and then along (0, 1)
        grid = np.fft.ifftshift(np.fft.ifftshift(np.fft.ifftshift(grid, axes=0), axes=0), axes=(0, 1))
        if half_scan_percentage != 0:
            grid[: int(np.round(n0 * half_scan_percentage)), :] = 0.0
        return _as_mask_tensor(grid[0], shape, [-2]), acceleration


class Gaussian2DMaskFunc(MaskFunc):
    """Points drawn without replacement from a separable Gaussian density (square root of the outer product of the two axis
    profiles), plus a fully sampled centre ellipse with half-axes `scale` x the extents (subsample.py:376-462)."""

    def __call__(self, shape, seed: Seed = None, half_scan_percentage=0.0, scale=0.02):
        self.shape = tuple(int(v) for v in shape[-3:-1])
        fwhm, acceleration = self.choose_acceleration()
        self.full_width_half_maximum = _fwhm_pair(fwhm)
        self.acceleration = acceleration
        self.scale = scale
        n0, n1 = self.shape
        xx, yy = np.mgrid[:n0, :n1]
        ellipse = np.power((xx - n0 / 2) / (scale * n0), 2) + np.power((yy - n1 / 2) / (scale * n1), 2)
        grid = (ellipse < 1).astype(float)
        density = np.sqrt(np.outer(_gaussian_profile(self.full_width_half_maximum[0], n0),
                                   _gaussian_profile(self.full_width_half_maximum[1], n1)))
        density = density / density.sum()
        flat = np.random.choice(range(n0 * n1), size=int(n0 * n1 / acceleration), replace=False, p=density.flatten())
        grid[np.unravel_index(flat, (n0, n1))] = 1.0
        if half_scan_percentage != 0:
            grid[: int(np.round(n0 * half_scan_percentage)), :] = 0.0
        return _as_mask_tensor(grid, shape, [-3, -2]), acceleration


class Poisson2DMaskFunc(MaskFunc):
    """Variable-density Poisson-disc masks (subsample.py:465-633; the algorithm is sigpy.mri.samp.poisson's).

    Samples keep an elliptical exclusion zone whose radius grows linearly with the normalised distance r from the (calibration region of
    the) k-space centre: radius = (1 + slope r) n_axis / max(n); the slope is found by bisection so that the achieved acceleration
    (grid size / number of samples) is within `tol` of the requested one.  The corners r >= 1 are dropped (`crop_corner`), a centred disc of
    radius `scale` x rows is fully sampled, `half_scan_percentage` zeroes the leading rows.  Returns (mask [1, ny, nx, 1] float32, acceleration).

    The reference throws its darts inside a Numba-compiled loop on Numba's own generator: its masks are not reproducible from `seed` even
    there.  Here the dart throwing is libmridc_amd's host routine `mrx_poisson_disc_mask` (the same loop on an explicit xoshiro256** stream),
    seeded from this object's `RandomState` under `temp_seed(seed)` like the other generators: the same seed gives the same mask.  Masks are
    therefore equal to the reference's in law (tests: acceleration, calibration box, centre disc, exclusion distances), not bit for bit."""

    def __call__(self, shape, seed: Seed = None, half_scan_percentage=0.0, scale=0.02, calib=(0.0, 0.0), crop_corner=True,
                 max_attempts=30, tol=0.3):
        from mridc_amd import _lib
        self.shape = tuple(int(v) for v in shape[-3:-1])
        self.scale = scale
        ny, nx = self.shape
        with temp_seed(self.rng, seed):
            _, self.acceleration = self.choose_acceleration()
            # normalised distance from the edge of the calibration block, per axis in [0, 1]
            rows, cols = np.mgrid[:ny, :nx]
            dx = np.maximum(np.abs(cols - nx / 2) - calib[-1] / 2, 0)
            dy = np.maximum(np.abs(rows - ny / 2) - calib[-2] / 2, 0)
            r = np.hypot(dx / dx.max(), dy / dy.max())
            longest = max(nx, ny)
            lo, hi = 0.0, float(longest)
            mask = achieved = None
            while lo < hi:
                slope = (lo + hi) / 2
                spread = 1 + r * slope
                rx = np.ascontiguousarray(np.clip(spread * nx / longest, 1, None), dtype=np.float32)
                ry = np.ascontiguousarray(np.clip(spread * ny / longest, 1, None), dtype=np.float32)
                stream = int(self.rng.randint(0, 2 ** 31 - 1)) * (2 ** 31) + int(self.rng.randint(0, 2 ** 31 - 1))
                cells = np.zeros((ny, nx), dtype=np.uint8)
                count = _lib.lib().mrx_poisson_disc_mask(nx, ny, int(max_attempts), rx.ctypes.data, ry.ctypes.data, float(calib[-1]),
                                                              float(calib[-2]), stream, cells.ctypes.data)
                if count < 0:
                    raise RuntimeError("mrx_poisson_disc_mask failed")
                mask = cells.astype(bool)
                if crop_corner:
                    mask &= r < 1
                kept = int(mask.sum())
                achieved = mask.size / kept if kept else np.inf
                if abs(achieved - self.acceleration) < tol:
                    break
                if achieved < self.acceleration:          # too dense: larger exclusion zones
                    lo = slope
                else:
                    hi = slope
        if abs(achieved - self.acceleration) >= tol:
            raise ValueError(f"Cannot generate mask to satisfy acceleration factor of {self.acceleration}.")
        mask = mask | self.centered_circle()
        if half_scan_percentage != 0:
            mask[: int(np.round(mask.shape[0] * half_scan_percentage)), :] = False
        return torch.from_numpy(mask.astype(np.float32)).unsqueeze(0).unsqueeze(-1), self.acceleration

    def centered_circle(self):
        """The fully sampled disc around the centre: radius int(rows x scale) pixels (subsample.py:539-546)."""
        n0, n1 = self.shape
        ii, jj = np.indices(self.shape)
        return (ii - int((n0 - 1) / 2)) ** 2 + (jj - int((n1 - 1) / 2)) ** 2 < int(n0 * self.scale) ** 2


def create_mask_for_mask_type(mask_type_str: str, center_fractions: Sequence[float], accelerations: Sequence[int]) -> MaskFunc:
    """subsample.py:636-664."""
    table = {"random1d": RandomMaskFunc, "equispaced1d": Equispaced1DMaskFunc, "equispaced2d": Equispaced2DMaskFunc,
             "gaussian1d": Gaussian1DMaskFunc, "gaussian2d": Gaussian2DMaskFunc, "poisson2d": Poisson2DMaskFunc}
    if mask_type_str not in table:
        raise NotImplementedError(f"{mask_type_str} not supported")
    return table[mask_type_str](center_fractions, accelerations)
