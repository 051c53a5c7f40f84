"""Worker of test_torch_tensors_in_and_out: the reference-shaped classes on torch tensors that live on the GPU.  torch first (its
HIP runtime has to be up before the engine binds the device), then the package."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (initialised before the engine binds the device)

assert torch.cuda.is_available()
torch.zeros(1, device="cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import xugrid_amd as xa  # noqa: E402
from xugrid_amd import meshgen  # noqa: E402


def same_or_nan(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def main():
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
    sxy, sf = meshgen.triangle_mesh(6000, 51)
    txy, tf = meshgen.triangle_mesh(5000, 52, 30.0, 0.8)
    src_h = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt_h = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    rng = np.random.default_rng(1)
    data = rng.normal(size=(3, sf.shape[0]))
    data[1, ::11] = np.nan
    src_d = xa.Ugrid2d.from_device_arrays(t(sxy), t(sf))
    tgt_d = xa.Ugrid2d.from_device_arrays(t(txy), t(tf))
    for cls, kwargs in ((xa.OverlapRegridder, {"method": "mean"}), (xa.RelativeOverlapRegridder, {}),
                        (xa.CentroidLocatorRegridder, {}), (xa.BarycentricInterpolator, {})):
        expected = cls(src_h, tgt_h, **kwargs).regrid(data)
        rg = cls(src_d, tgt_d, **kwargs)
        # the data produced by a torch kernel right before the call (the engine waits for torch's stream)
        d = t(data) * 2.0
        got = rg.regrid(d)
        assert isinstance(got, torch.Tensor) and got.is_cuda and got.dtype == torch.float64 and tuple(got.shape) == expected.shape
        exp2 = cls(src_h, tgt_h, **kwargs).regrid(data * 2.0)
        assert same_or_nan(got.cpu().numpy(), exp2).all(), cls.__name__
        got32 = rg.regrid(t(data.astype(np.float32)))
        assert same_or_nan(got32.cpu().numpy(), cls(src_h, tgt_h, **kwargs).regrid(data.astype(np.float32))).all()
        # torch goes on computing with the result
        assert torch.isfinite(torch.nansum(got)).item()
        # non-contiguous tensors are refused, host tensors take the host path's error
        try:
            rg.regrid(t(np.ascontiguousarray(data.T)).T)
            raise SystemExit("a non-contiguous tensor was accepted")
        except ValueError:
            pass
        try:
            rg.regrid(torch.as_tensor(data))
            raise SystemExit("a CPU tensor was accepted")
        except TypeError:
            pass
    print("TORCH_DEVICE_API_OK")


if __name__ == "__main__":
    main()
