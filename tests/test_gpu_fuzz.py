"""A bounded run of the randomised shape sweep (tests/fuzz_gpu.py): 160 drawn cases: codePattern, warp_p2c,
bilinear_sampler and coord_transform against the CPU oracle, the capture and re-projection kernels against the
composed operators, the sampler's image gradient against a float64 scatter."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_against_the_oracle(seed):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import fuzz_gpu
    failures = fuzz_gpu.sweep(80, seed)
    assert not failures, failures
