"""Three launches of supir_flash_attn_d512 at T = 16 384 (the VAE mid block at 1024^2 px) for a rocprofv3 --pmc pass.
  cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- python tools/pmc_d512.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from supir_amd import ops  # noqa: E402

T = 16384
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(1, T, 512, generator=g).to("cuda", torch.bfloat16) for _ in range(3))
vt = v.permute(0, 2, 1).contiguous()
for _ in range(3):
    ops.flash_attn_d512(q, k, vt, T)
torch.cuda.synchronize()
