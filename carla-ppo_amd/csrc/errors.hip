// errors.hip — last-error string of the C ABI (thread-local; the library keeps no other global state).
#include "mi_internal.hpp"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int mi_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

int mi_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MI_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return MI_ERR_LAUNCH;
}

extern "C" const char* mi_last_error(void) { return g_err; }

thread_local hipEvent_t mi_tl_stop_event = nullptr;     // mi_internal.hpp, MI_LAUNCH

extern "C" int mi_abi_version(void) { return 7; }          // 2: frames_u8 arguments, engine-side noise, mi_vae_train_step; 3: split storage (MI_BF16X3), fused decoder tail, mi_ppo_train_step_idx, mi_comm_probe; 4: ordered (atomic-free) gradient reductions: mi_colsum_ws / mi_gemm_wgrad_ws, mi_ppo_fused_shape_ok; 5: MlpVAE engine (mi_mlpvae_*), mi_adam_tf_layouts, mi_gather_rows_cast, mi_gemm_wgrad_bias_set; 6: mi_device_probe (box calibration), mi_vae_train_step_dp, mi_vae_reparam_kl_fwd_bwd, frames_u8 on the mi_mlpvae_* entry points, MiVaeDesc.inference_only; 7: mi_ppo_train_step_dp, mi_mlpvae_train_step_dp / mi_mlpvae_dp_buckets (one-call data-parallel steps of PPO and the MlpVAE)

// device properties the host side reports in bench output (no torch needed)
extern "C" int mi_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return mi_fail(MI_ERR_STATE, "hipGetDeviceProperties failed");
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    return MI_OK;
}

// CRC-32C (Castagnoli) of a host buffer, slicing-by-8: the checksum of TensorFlow's tensor-bundle checkpoints (every tensor in the
// .data shard and every block of the .index table carries one, tensorflow/core/lib/hash/crc32c.h).  `crc` = running value (0 to start).
// Host-side I/O helper of the checkpoint reader / writer (mi355/tf_bundle.py); no device work.
static uint32_t g_crc_tab[8][256];
static bool g_crc_init = false;
static void crc32c_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
    g_crc_init = true;
}
extern "C" unsigned int mi_crc32c(unsigned int crc, const void* data, long long n) {
    if (!g_crc_init) crc32c_init();                      // idempotent: a racing second initialiser writes the same values
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~crc;
    while (n > 0 && (((uintptr_t)p) & 7)) { c = g_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = g_crc_tab[7][w & 0xff] ^ g_crc_tab[6][(w >> 8) & 0xff] ^ g_crc_tab[5][(w >> 16) & 0xff] ^ g_crc_tab[4][(w >> 24) & 0xff] ^
            g_crc_tab[3][(w >> 32) & 0xff] ^ g_crc_tab[2][(w >> 40) & 0xff] ^ g_crc_tab[1][(w >> 48) & 0xff] ^ g_crc_tab[0][(w >> 56) & 0xff];
        p += 8; n -= 8;
    }
    while (n-- > 0) c = g_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}
