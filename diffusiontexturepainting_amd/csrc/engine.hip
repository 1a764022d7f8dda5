// placeholder while the engine is being written (replaced in the next commit)
#include "../../include/dtp.h"
#include "common.h"
extern "C" {
#define NI { dtp_set_error("not implemented yet"); return DTP_ERR_STATE; }
int dtp_create(int, int, int, dtp_ctx**) NI
void dtp_destroy(dtp_ctx*) {}
int dtp_load_tensor(dtp_ctx*, const char*, const float*, int, const int64_t*, int) NI
int dtp_finalize_weights(dtp_ctx*) NI
int dtp_vae_encode(dtp_ctx*, const float*, const float*, float*, int, dtp_stream) NI
int dtp_unet(dtp_ctx*, const float*, float, const void*, float*, int, dtp_stream) NI
int dtp_vae_decode(dtp_ctx*, const float*, float*, int, dtp_stream) NI
int dtp_set_brush(dtp_ctx*, const float*, int, int, float*, dtp_stream) NI
int dtp_set_conditioning(dtp_ctx*, const float*, const float*, const float*, dtp_stream) NI
int dtp_get_conditioning(dtp_ctx*, float*, float*, dtp_stream) NI
int dtp_stamp(dtp_ctx*, const float*, const dtp_settings*, const float*, const float*, void*, int, dtp_stream) NI
int dtp_last_stamp_times(dtp_ctx*, float*) NI
int dtp_last_stamp_info(dtp_ctx*, int*, int*) NI
}
