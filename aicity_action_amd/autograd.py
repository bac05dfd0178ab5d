"""Training path (forward with saved activations + hand-written backward kernels).

Round-1 status: the backward kernels are not built yet, so asking for gradients fails loudly
instead of silently falling back to eager PyTorch.
"""


def forward_with_grad(model, clip, return_logits=False):
    raise NotImplementedError(
        "MViT (HIP path): backward kernels are not built yet; run forward under torch.no_grad() "
        "(inference / parity / forward benchmark). No eager fallback is provided on purpose.")
