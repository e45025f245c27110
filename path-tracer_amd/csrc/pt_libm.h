// Device libm used by the integrator (see pt_math.h).
#pragma once

PT_D float pt_acosf(float x) { return acosf(x); }
PT_D float pt_sinf(float x) { return sinf(x); }
PT_D float pt_cosf(float x) { return cosf(x); }
// powf(x, 1/2.2f) of Renderer::post_processing (renderer/mod.rs:339-345)
PT_D float pt_pow_inv_gamma(float x) { return powf(x, 1.0f / 2.2f); }
