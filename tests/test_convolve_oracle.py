"""`convolve` (wasm/Convolve.h:23-92): the CPU restatement of the two-stage FFT convolver
(oracle/fftconv_oracle.h) against outputs recorded from the reference's own prebuilt wasm engine
(tests/golden/convolve_wasm.f32), and both against float64 linear convolution."""
import numpy as np
import pytest

import conv_cases as C
import oracle

TOL = 1e-6


@pytest.mark.parametrize("name", sorted(C.SCENARIOS))
def test_golden_is_a_linear_convolution(name):
    """Pins the scenario generators (IRs, input stream, swap/reset semantics) to the recording."""
    assert float(np.abs(C.golden(name).astype(np.float64) - C.exact_model(name)).max()) <= 2e-7


@pytest.mark.skipif(not oracle.have_port(), reason="oracle port not built")
@pytest.mark.parametrize("name", sorted(C.SCENARIOS))
def test_restatement_matches_reference_wasm(name):
    y = C.run_scenario(lambda sr, bs: oracle.PortRuntime(sr, bs), name)
    g = C.golden(name)
    assert y.shape == g.shape
    assert float(np.abs(y.astype(np.float64) - g).max()) <= TOL


@pytest.mark.skipif(not oracle.have_port(), reason="oracle port not built")
def test_convolve_property_codes():
    rt = oracle.PortRuntime(48000.0, 512)
    assert rt.apply_instructions([[0, 1, "convolve"], [3, 1, "path", 4]]) == 5          # Convolve.h:37-38
    assert rt.apply_instructions([[3, 1, "path", "/nope"]]) == 6                         # :40-41
