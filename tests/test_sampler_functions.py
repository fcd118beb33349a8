"""sample_pdf(det=True) and NeuS.up_sample on their own, through the C ABI (cnr_sample_pdf / cnr_up_sample), against the golden vectors
captured from the reference (ray_utils.py:123-154, NeuS.py:136-181): generic rows plus the edge cases the reference's semantics
define -- all-zero weights (pdf only from the +1e-5), flat weights (ties in the cdf), a single spike (long flat cdf runs), leading
zeros, tiny weights (the ``denom < 1e-5 -> 1`` switch).  The CPU-emulation build runs here, the HIP upsample_kernel under -m gpu."""
import os

import numpy as np
import pytest
import torch

import _golden as G
import _native as N
import color_neus_amd as cn


def _sample_pdf(library, device):
    fx = G.load("functions")
    bins, w = torch.from_numpy(fx["spdf:bins"]).to(device), torch.from_numpy(fx["spdf:w"]).to(device)
    for m in (16, 5):
        got = cn.sample_pdf(bins, w, m, det=True, library=library).cpu().numpy()
        ref = fx[f"spdf:out{m}"]
        assert got.shape == ref.shape
        # positions inside [1, 3]; identical op order, so agreement is at float32 round-off of the cdf (1e-6), far below a bin width
        err = np.abs(got - ref).max(axis=1)
        assert (err < 2e-5).all(), (m, err)
        assert (np.diff(got, axis=1) >= 0).all(), "samples must be monotone"


def _sample_pdf_random(library, device):
    """det=False (ray_utils.py:135-136): u = torch.rand on the CPU generator with the reference's shape -- the same seed gives the reference's
    draws, hence its samples (golden: tools/gen_golden.py --spdf-random-only)."""
    fx = G.load("sample_pdf_random")
    bins, w = torch.from_numpy(fx["bins"]).to(device), torch.from_numpy(fx["w"]).to(device)
    for m in (16, 5):
        torch.manual_seed(int(fx["seed"]))
        got = cn.sample_pdf(bins, w, m, det=False, library=library).cpu().numpy()
        assert np.abs(got - fx[f"out{m}"]).max() < 2e-5, m
        # and the generator was consumed exactly like the reference consumes it
        torch.manual_seed(int(fx["seed"]))
        torch.rand([bins.shape[0], m])
        nxt = torch.rand(1)
        torch.manual_seed(int(fx["seed"]))
        dflt = cn.sample_pdf(bins, w, m, library=library)   # the DEFAULT call is the reference's default: det=False (ray_utils.py:121)
        assert torch.equal(torch.rand(1), nxt)
        assert np.array_equal(dflt.cpu().numpy(), got)


def _up_sample(library, device):
    fx = G.load("functions")
    from oracle import colorneus_oracle as O
    r = N.make_renderer(O.tiny_config(), G.prefixed(fx, "tinyw:"), library, device)
    t = lambda k: torch.from_numpy(fx[k]).to(device)
    for i in range(4):
        got = r.up_sample(t("ups:o"), t("ups:d"), t("ups:z"), t("ups:sdf"), 4, 64 * 2 ** i).cpu().numpy()
        assert np.abs(got - fx[f"ups:new_z_{i}"]).max() < 1e-4, i   # the z range is ~2; a coarse section is 0.13 long


def _crafted_rows(library, device):
    """Rows built to sit exactly on the switches: duplicate bins, weights that make consecutive cdf entries equal, u landing exactly on
    a cdf entry (searchsorted right=True), denom just below / above 1e-5.  Checked against the oracle's restatement of sample_pdf."""
    from oracle import colorneus_oracle as O
    n = 9
    bins = torch.linspace(1.0, 3.0, n).repeat(6, 1)
    bins[1, 3] = bins[1, 4]                                  # duplicate bin edge
    w = torch.zeros(6, n - 1)
    w[0] = 1.0                                               # u_k = (k + .5)/4 with 8 equal sections: u hits cdf entries exactly
    w[1] = torch.tensor([0, 0, 1, 1, 0, 0, 1, 0.0])
    w[2, -1] = 1.0                                           # everything in the last section
    w[3, 0] = 1.0                                            # everything in the first section
    w[4] = torch.tensor([1e-5, 0, 0, 0, 0, 0, 0, 1e-5])      # cdf jumps comparable to the 1e-5 switch
    w[5] = 3e-6                                              # all sections below the switch
    for m in (4, 16, 64):
        got = cn.sample_pdf(bins.to(device), w.to(device), m, det=True, library=library).cpu()
        ref = O.sample_pdf_det(bins, w, m)
        assert float((got - ref).abs().max()) < 2e-5, m


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_sample_pdf_emu():
    _sample_pdf(N.EMU_LIB, "cpu")
    _crafted_rows(N.EMU_LIB, "cpu")
    _sample_pdf_random(N.EMU_LIB, "cpu")


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_up_sample_emu():
    _up_sample(N.EMU_LIB, "cpu")


@pytest.mark.gpu
def test_sample_pdf_hip():
    _sample_pdf(None, "cuda:0")
    _crafted_rows(None, "cuda:0")
    _sample_pdf_random(None, "cuda:0")


@pytest.mark.gpu
def test_up_sample_hip():
    _up_sample(None, "cuda:0")


def test_argument_checks():
    lib = cn.load_library(N.EMU_LIB) if os.path.isfile(N.EMU_LIB) else None
    if lib is None:
        pytest.skip("emulation library not built")
    with pytest.raises(RuntimeError):
        cn.sample_pdf(torch.zeros(2, 300), torch.zeros(2, 299), 16, library=lib)      # more than 256 bins
    with pytest.raises(RuntimeError):
        cn.sample_pdf(torch.zeros(2, 8), torch.zeros(2, 7), 65, library=lib)          # more than 64 samples
    assert cn.sample_pdf(torch.linspace(0, 1, 8).repeat(2, 1), torch.ones(2, 7), 4, det=False, library=lib).shape == (2, 4)
