"""tests/torch_ops.py is a scoped stand-in: outside `installed()` supir_amd.ops is the HIP path and refuses CPU tensors."""
import pytest
import torch

from tests import torch_ops


def test_backend_restores_the_real_ops_on_exit():
    """The stand-in is scoped: outside `installed()` supir_amd.ops is the HIP path again and still refuses CPU tensors."""
    from supir_amd import _lib, ops
    from supir_amd import weights as Wt
    with torch_ops.installed(fp32=True):
        assert ops.gemm is torch_ops.gemm and Wt.cdt() == torch.float32
    assert ops.gemm is not torch_ops.gemm and Wt.cdt() == torch.bfloat16
    with pytest.raises(_lib.SupirHipError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))
