"""Box calibration through the C ABI (round 5): mi_device_probe reports rates a MI355X can have, and bench.py's `box` object carries them."""
import numpy as np
import pytest
import torch

from mi355 import lib as milib

pytestmark = pytest.mark.gpu


def test_device_probe_reports_plausible_rates():
    L = milib.get()
    nbytes = 512 << 20
    scratch = torch.empty(nbytes, device="cuda", dtype=torch.uint8)
    out8 = np.zeros(8, np.float32)
    torch.cuda.synchronize()
    L.mi_device_probe(torch.cuda.current_stream().cuda_stream, scratch.data_ptr(), nbytes, 2, out8.ctypes.data)
    mfma, sclk, mfma0, sclk0, rd, cp, cus, walked = [float(x) for x in out8]
    assert cus == torch.cuda.get_device_properties(0).multi_processor_count
    assert 500.0 < mfma <= 2600.0 and 500.0 < mfma0 <= 2600.0, (mfma, mfma0)          # dense bf16 peak 2.5 PF at 2.4 GHz
    assert 800.0 < sclk <= 2500.0 and 800.0 < sclk0 <= 2500.0, (sclk, sclk0)
    # the MFMA rate IS the clock: one 32x32x16 MFMA per SIMD per 32 cycles = 4096 FLOP / cycle / CU.  (The clock is one wave's s_memtime / s_memrealtime over its own
    # life, the rate is work / the launch's event time -- launch + drain and a clock that moves during a 2 ms launch keep the two a few per cent apart: 5.2 % seen.)
    assert abs(mfma * 1e12 / (cus * 4096.0 * sclk * 1e6) - 1.0) < 0.12, (mfma, sclk)
    assert 1.0 < rd < 8.5 and 1.0 < cp < 8.5, (rd, cp)
    assert walked > 0.9 * (nbytes - 65536)
