// Brush (condition-patch) encoder: placeholder until the ViT path lands (next commit).
#include "engine.h"
int load_imgenc_weights(Ctx* c) { c->ienc.present = false; return DTP_OK; }
extern "C" int dtp_set_brush(dtp_ctx*, const float*, int, int, float*, dtp_stream) {
  dtp_set_error("dtp_set_brush: image-encoder weights (clip.*, penc.*) were not loaded");
  return DTP_ERR_STATE;
}
