// Beam search and general-length additive attention (filled in below in later commits).
#include "common.h"

extern "C" size_t i2l_beam_workspace_bytes(int images, int beam, int hidden, int layers, int steps) {
    (void)images; (void)beam; (void)hidden; (void)layers; (void)steps;
    return 0;
}

extern "C" int i2l_beam_decode(const i2l_decoder_weights* w, const void* workspace, int images, int beam, int steps,
                               int start_id, int end_id, void* beam_workspace, size_t beam_workspace_bytes,
                               int32_t* seq_out, int32_t* len_out, double* score_out, i2l_stream_t stream) {
    (void)w; (void)workspace; (void)images; (void)beam; (void)steps; (void)start_id; (void)end_id;
    (void)beam_workspace; (void)beam_workspace_bytes; (void)seq_out; (void)len_out; (void)score_out; (void)stream;
    return I2L_ERR_UNSUPPORTED;
}

extern "C" int i2l_attention_context_fwd(const float* hidden, const float* enc, const float* w_attn,
                                         const float* b_attn, const float* v, float* context, int B, int S, int H,
                                         int E, i2l_stream_t stream) {
    (void)hidden; (void)enc; (void)w_attn; (void)b_attn; (void)v; (void)context; (void)B; (void)S; (void)H; (void)E;
    (void)stream;
    return I2L_ERR_UNSUPPORTED;
}
