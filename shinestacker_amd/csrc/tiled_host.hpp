// tiled_host.hpp -- host side of the tiled path and of host-frame staging.
// Included by capi.hip after `struct mi_stack` is complete.
#pragma once

namespace mi {

bool tiled_available() { return false; }
int tiled_create(mi_stack*) { return MI_OK; }
void tiled_destroy(mi_stack*) {}
int tiled_reset(mi_stack*) { return MI_OK; }
int tiled_pending(const mi_stack*) { return 0; }
int tiled_flush(mi_stack*) { return MI_OK; }
int tiled_push(mi_stack*, const void*, int, size_t) {
    return fail(MI_ERR_UNSUPPORTED, "tiled implementation not built");
}
const float* tiled_last_gauss(mi_stack* s, int level) { return s->G[level]; }

// One host frame: stage into device memory, then run the device path.
int tiled_push_host(mi_stack* s, const void* host_bgr, size_t row_stride_bytes) {
    const size_t esz = s->p.in_dtype == MI_U8 ? 1 : (s->p.in_dtype == MI_U16 ? 2 : 4);
    const size_t rb = (size_t)s->p.width * 3 * esz;
    if (row_stride_bytes == 0) row_stride_bytes = rb;
    if (row_stride_bytes < rb) return fail(MI_ERR_INVALID, "row stride smaller than a row");
    MI_HIP(hipMemcpy2DAsync(s->frame_dev, rb, host_bgr, row_stride_bytes, rb, s->p.height,
                            hipMemcpyHostToDevice, s->stream));
    // pageable source: make sure the host buffer is free to reuse on return
    MI_HIP(hipStreamSynchronize(s->stream));
    return dispatch_push(s, s->frame_dev, 1, rb * s->p.height);
}

}  // namespace mi
