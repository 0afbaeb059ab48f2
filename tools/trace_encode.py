"""list the kernel launches of ONE eager encode / decode step in dispatch order (run under rocprofv3 --kernel-trace; tools/_bin scripts
print the tail of the trace CSV).  Used to find ATen kernels that sneak into the captured hot path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_video  # noqa: E402

torch.set_grad_enabled(False)
what = sys.argv[1] if len(sys.argv) > 1 else 'encode'
cv, _, _, _ = build_models('bf16', False)
video = synthetic_video(8, 17, 256, 0).cuda()
ids = cv(video, return_only_codebook_ids=True)
cv.decode_from_codebook_indices(ids)                              # (packs the decoder's weights)
torch.cuda.synchronize()
marker = torch.zeros(7, device='cuda', dtype=torch.float64)       # an unmistakable FillFunctor<double> in the trace
marker.fill_(1.0)
if what == 'encode':
    cv(video, return_only_codebook_ids=True)
else:
    cv.decode_from_codebook_indices(ids)
marker.fill_(2.0)
torch.cuda.synchronize()
