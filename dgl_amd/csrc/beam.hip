#include "common.h"
extern "C" int dgl_beam_advance(const float *, float *, int32_t, int32_t, int32_t, int32_t, int32_t *, int32_t *, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
extern "C" int dgl_beam_advance_compact(const dgl_policy_io_t *, const float *, float *, int32_t, int32_t, int32_t, int32_t, int32_t *, int32_t *, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
