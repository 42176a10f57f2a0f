// encoder.hip -- BERT-large encoder entry points (ollama_embed_text stand-in, main.py:134-145).
// Placeholder until the HIP encoder lands: every entry point reports UNSUPPORTED loudly.
#include "kernels.h"

using namespace sqe;

extern "C" {

int sqe_encoder_create(sqe_ctx*, const sqe_bert_cfg*, sqe_encoder** out) {
    if (out) *out = nullptr;
    return fail(SQE_ERR_UNSUPPORTED, "encoder: not implemented in this build");
}
void sqe_encoder_destroy(sqe_encoder*) {}
int sqe_encoder_load_tensor(sqe_encoder*, const char*, const float*, const int64_t*, int) {
    return fail(SQE_ERR_UNSUPPORTED, "encoder: not implemented in this build");
}
int sqe_encoder_finalize(sqe_encoder*) { return fail(SQE_ERR_UNSUPPORTED, "encoder: not implemented in this build"); }
int sqe_encode(sqe_encoder*, const int32_t*, const int32_t*, int, int, float*) {
    return fail(SQE_ERR_UNSUPPORTED, "encoder: not implemented in this build");
}
int sqe_encode_device(sqe_encoder*, const int32_t*, const int32_t*, int, int, float*) {
    return fail(SQE_ERR_UNSUPPORTED, "encoder: not implemented in this build");
}

}  // extern "C"
