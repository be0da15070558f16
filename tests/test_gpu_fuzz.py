"""A bounded run of the randomised shape sweep (tests/fuzz_gpu.py): 128 drawn cases of codePattern and
warp_p2c against the CPU oracle and of the capture and re-projection kernels against the composed operators."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_against_the_oracle(seed):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import fuzz_gpu
    failures = fuzz_gpu.sweep(64, seed)
    assert not failures, failures
