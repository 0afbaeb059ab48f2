"""list the kernel launches of ONE eager encode / decode step (or one whole `sample` call) in dispatch order (run under rocprofv3 --kernel-trace; tools/_bin scripts
print the tail of the trace CSV).  Used to find ATen kernels that sneak into the captured hot path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_video  # noqa: E402

torch.set_grad_enabled(False)
what = sys.argv[1] if len(sys.argv) > 1 else 'encode'
cv, _, _, ph = build_models(os.environ.get('PK_TRACE_DTYPE', 'bf16'), what == 'sample')
video = synthetic_video(8, 17, 256, 0).cuda()
ids = cv(video, return_only_codebook_ids=True)
cv.decode_from_codebook_indices(ids)                              # (packs the decoder's weights)
torch.cuda.synchronize()
marker = torch.zeros(7, device='cuda', dtype=torch.float64)       # an unmistakable FillFunctor<double> in the trace
marker.fill_(1.0)
if what == 'sample':
    from bench import synthetic_context
    ctx = synthetic_context(8, 12, 768, seed=1).cuda()
    ph.encode_texts = lambda texts, output_device=None: ctx[:len(texts)]
    for _ in range(2):
        ph.sample(texts=['x'] * 8, num_frames=17, cond_scale=5.)
    torch.cuda.synchronize()
    marker.fill_(1.0)
    ph.sample(texts=['x'] * 8, num_frames=17, cond_scale=5.)
elif what == 'encode':
    cv(video, return_only_codebook_ids=True)
else:
    cv.decode_from_codebook_indices(ids)
marker.fill_(2.0)
torch.cuda.synchronize()
