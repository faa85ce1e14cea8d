/* What the card reports about itself while the file path runs: measurement infrastructure of bench.py and tools/clock_probe.sh,
 * not part of the product library.  One sample = the gpu_metrics table of device 0 through ROCm SMI (the struct layout is the
 * header's, which is why this is C and not ctypes).  Built by __graft_entry__.build():
 *   gcc -O2 -shared -fPIC -I/opt/rocm/include tools/smi/smi_probe.c -L/opt/rocm/lib -lrocm_smi64 -o tools/smi/libsmi_probe.so */
#include <stdint.h>
#include <string.h>
#include "rocm_smi/rocm_smi.h"

enum { SMI_GFX_MEAN, SMI_GFX_MIN, SMI_GFX_MAX, SMI_UCLK, SMI_POWER_W, SMI_T_HOTSPOT, SMI_T_MEM, SMI_ACC_COUNTER, SMI_PPT_ACC,
       SMI_SOCKET_THM_ACC, SMI_HBM_THM_ACC, SMI_PROCHOT_ACC, SMI_VR_THM_ACC, SMI_POWER_CAP_W, SMI_GFX_ACTIVITY, SMI_UMC_ACTIVITY,
       SMI_NFIELDS };

static int g_open = 0;

int smi_nfields(void) { return SMI_NFIELDS; }

int smi_open(void) {
    if (g_open) return 0;
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) return -1;
    uint32_t n = 0;
    if (rsmi_num_monitor_devices(&n) != RSMI_STATUS_SUCCESS || n == 0) return -2;
    g_open = 1;
    return 0;
}

/* out[SMI_NFIELDS]; a field the table does not carry is -1.  Returns 0, or the rsmi status of the metrics call. */
int smi_sample(uint32_t dev, double *out) {
    for (int i = 0; i < SMI_NFIELDS; ++i) out[i] = -1.0;
    if (!g_open) return -1;
    rsmi_gpu_metrics_t m;
    memset(&m, 0xff, sizeof m);
    rsmi_status_t rc = rsmi_dev_gpu_metrics_info_get(dev, &m);
    if (rc != RSMI_STATUS_SUCCESS) return (int)rc;
    double s = 0.0, lo = 1e9, hi = -1.0; int n = 0;
    for (int i = 0; i < RSMI_MAX_NUM_GFX_CLKS; ++i) {
        const uint16_t c = m.current_gfxclks[i];
        if (c == 0xffff || c == 0) continue;
        s += c; ++n; if (c < lo) lo = c; if (c > hi) hi = c;
    }
    if (n) { out[SMI_GFX_MEAN] = s / n; out[SMI_GFX_MIN] = lo; out[SMI_GFX_MAX] = hi; }
    else if (m.current_gfxclk != 0xffff) out[SMI_GFX_MEAN] = out[SMI_GFX_MIN] = out[SMI_GFX_MAX] = m.current_gfxclk;
    if (m.current_uclk != 0xffff) out[SMI_UCLK] = m.current_uclk;
    if (m.current_socket_power != 0xffff) out[SMI_POWER_W] = m.current_socket_power;
    else if (m.average_socket_power != 0xffff) out[SMI_POWER_W] = m.average_socket_power;
    if (m.temperature_hotspot != 0xffff) out[SMI_T_HOTSPOT] = m.temperature_hotspot;
    if (m.temperature_mem != 0xffff) out[SMI_T_MEM] = m.temperature_mem;
    if (m.accumulation_counter != UINT64_MAX) out[SMI_ACC_COUNTER] = (double)m.accumulation_counter;
    if (m.ppt_residency_acc != UINT64_MAX) out[SMI_PPT_ACC] = (double)m.ppt_residency_acc;
    if (m.socket_thm_residency_acc != UINT64_MAX) out[SMI_SOCKET_THM_ACC] = (double)m.socket_thm_residency_acc;
    if (m.hbm_thm_residency_acc != UINT64_MAX) out[SMI_HBM_THM_ACC] = (double)m.hbm_thm_residency_acc;
    if (m.prochot_residency_acc != UINT64_MAX) out[SMI_PROCHOT_ACC] = (double)m.prochot_residency_acc;
    if (m.vr_thm_residency_acc != UINT64_MAX) out[SMI_VR_THM_ACC] = (double)m.vr_thm_residency_acc;
    if (m.average_gfx_activity != 0xffff) out[SMI_GFX_ACTIVITY] = m.average_gfx_activity;
    if (m.average_umc_activity != 0xffff) out[SMI_UMC_ACTIVITY] = m.average_umc_activity;
    uint64_t cap = 0;
    if (rsmi_dev_power_cap_get(dev, 0, &cap) == RSMI_STATUS_SUCCESS) out[SMI_POWER_CAP_W] = (double)cap / 1e6;
    return 0;
}
