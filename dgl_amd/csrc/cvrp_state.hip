#include "common.h"
extern "C" int dgl_cvrp_reset(const dgl_cvrp_state_t *, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
extern "C" int dgl_cvrp_step(const dgl_cvrp_state_t *, const int32_t *, int32_t, int32_t, int32_t, int32_t *, int32_t *, uint8_t *, float *, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
extern "C" int dgl_cvrp_state_data(const dgl_cvrp_state_t *, int32_t, float *, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
extern "C" int dgl_cvrp_route_length(const dgl_cvrp_state_t *, float *, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
extern "C" int dgl_cvrp_permute(const dgl_cvrp_state_t *, const dgl_cvrp_state_t *, const int32_t *, int32_t, int32_t, dgl_stream_t) { return dgl_set_error(DGL_ERR_UNSUPPORTED, "todo"); }
