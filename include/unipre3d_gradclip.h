/*
 * unipre3d_gradclip.h -- C-ABI of the gradient validity check + global-norm clip of UniPre3D's training loop
 * (second half).
 *
 * Replaces Trainer._check_and_clip_gradients (train_network.py:368-390):
 *     has_invalid = any(torch.isnan(p.grad).any() or torch.isinf(p.grad).any() for p in parameters)   # :376-380
 *     if has_invalid: return False                                                                     # caller skips optimizer.step(), :336-340
 *     torch.nn.utils.clip_grad_norm_(parameters, max_norm=1.0); return True                           # :386-389
 * i.e. two device->host synchronisations per parameter tensor (a few hundred per step) and clip_grad_norm_'s ~10 launches become
 * ONE multi-tensor pass over a DEVICE table of gradient pointers, a one-workgroup finalize and (only when the norm exceeds
 * max_norm) one multi-tensor scale pass; the caller reads ONE 32-byte state block -- or none at all: `state` also holds the
 * `grad_scale` / `found_inf` scalars torch's fused AdamW consumes on the device.
 *
 * Tensors: fp32, each contiguous; any 4-byte alignment (DDP's gradient_as_bucket_view hands out views at arbitrary offsets).
 * A tensor is cut into chunks of U3D_GC_CHUNK elements; one workgroup handles one chunk:
 *     first_chunk[t] = sum over s < t of ceil(numel[s] / U3D_GC_CHUNK)   (n_tensors + 1 entries, last = n_chunks)
 * Nothing here allocates or synchronises; all pointers are DEVICE pointers except where noted; kernels run on `stream`.
 * Returns 0 ok, 1 invalid argument, 3 launch failure.
 *
 * state (32 bytes, written by u3d_gradclip_finalize):
 *     double total_norm   global L2 norm: sqrt of the f64 sum of squares in a FIXED order (deterministic; cannot overflow)
 *     double amax         largest finite |g|
 *     float  coef         clip_grad_norm_'s coefficient min(1, max_norm / (total_norm + 1e-6)); 1 when a non-finite value was found
 *     float  grad_scale   1 / coef          (fused AdamW divides gradients by it)
 *     float  found_inf    1.0 if any gradient element is NaN or +-Inf, else 0.0   (fused AdamW skips the step when it is 1)
 *     float  reserved
 */
#ifndef UNIPRE3D_GRADCLIP_H
#define UNIPRE3D_GRADCLIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define U3D_GC_CHUNK 65536
#define U3D_GC_STATE_BYTES 32
#define U3D_GC_PARTIAL_BYTES 16 /* per chunk: double sum of squares, float max-abs, uint32 non-finite flag */

/* pass 1: per-chunk sum of squares (f64), max |g| over finite values, non-finite flag -> partials[n_chunks] */
int u3d_gradclip_stats(const void* const* grad_ptrs, const int64_t* numel, const int32_t* first_chunk, int32_t n_tensors,
                       int32_t n_chunks, void* partials, void* stream);
/* one workgroup: fixed-order combination of the partials -> state */
int u3d_gradclip_finalize(const void* partials, int32_t n_chunks, float max_norm, void* state, void* stream);
/* pass 2: g *= state.coef for every tensor, in place; every workgroup leaves at once when coef == 1 or found_inf == 1 */
int u3d_gradclip_scale(void* const* grad_ptrs, const int64_t* numel, const int32_t* first_chunk, int32_t n_tensors,
                       int32_t n_chunks, const void* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif
