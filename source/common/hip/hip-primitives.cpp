/* hip-primitives.cpp -- the file a maintainer drops into x265's source/common/hip/.
 *
 * Setup_HIP_Primitives() has the shape of Setup_Assembly_Primitives() (source/common/primitives.h:291-293
 * of the reference): it overwrites the ME cost-path entries of the function-pointer table with the
 * MI355X implementations exported by libx265_hip.so (include/x265_hip.h).  Call it from
 * x265_setup_primitives() after Setup_Assembly_Primitives() and before Setup_Alias_Primitives()
 * (source/common/primitives.cpp:134-143) so that the sad_square[] aliases pick the new entries up.
 *
 * Built only with -DENABLE_HIP=1 (see CMakeLists.txt next to this file).  It is compile-checked
 * against the reference's headers by `make -C oracle glue-check` where /root/reference exists.
 */
#include "primitives.h"
#include "x265.h"

#include "x265_hip.h"

namespace x265 {
// x265 private namespace

void Setup_HIP_Primitives(EncoderPrimitives &p, x265_param *param, int device)
{
    x265_hip_primitives t;
    int rc = x265_hip_setup_primitives(&t, device);

    if (rc != X265_HIP_OK)
    {
        /* keep the C / assembly entries, as the reference does when a faster path is unavailable
         * (primitives.cpp:137-141) */
        x265_log(param, X265_LOG_WARNING, "HIP primitives not available (%s), keeping CPU primitives\n",
                 x265_hip_last_error());
        return;
    }

    /* the two tables use the same partition index order (primitives.h:38-46) */
    for (int i = 0; i < NUM_LUMA_PARTITIONS; i++)
    {
        p.sad[i]      = (pixelcmp_t)t.sad[i];
        p.sad_x3[i]   = (pixelcmp_x3_t)t.sad_x3[i];
        p.sad_x4[i]   = (pixelcmp_x4_t)t.sad_x4[i];
        p.satd[i]     = (pixelcmp_t)t.satd[i];
        p.luma_hpp[i] = (filter_pp_t)t.luma_hpp[i];
        p.luma_hps[i] = (filter_hps_t)t.luma_hps[i];
        p.luma_vpp[i] = (filter_pp_t)t.luma_vpp[i];
        p.luma_vsp[i] = (filter_sp_t)t.luma_vsp[i];
        p.pixelavg_pp[i] = (pixelavg_pp_t)t.pixelavg_pp[i];
        p.sa8d_inter[i] = (pixelcmp_t)t.sa8d_inter[i];
        p.sse_pp[i] = (pixelcmp_t)t.sse_pp[i];
        p.luma_vps[i] = (filter_ps_t)t.luma_vps[i];
        p.luma_vss[i] = (filter_ss_t)t.luma_vss[i];
        p.luma_hvpp[i] = (filter_hv_pp_t)t.luma_hvpp[i];
        p.luma_addAvg[i] = (addAvg_t)t.luma_addAvg[i];
    }
    /* the plane producers either side of the path (primitives.h:232, 257) */
    p.luma_p2s = (filter_p2s_t)t.luma_p2s;
    p.weight_pp = (weightp_pp_t)t.weight_pp;
    p.frame_init_lowres_core = (downscale_t)t.frame_init_lowres_core;
    /* the margins of finished rows (primitives.h:259; reference.cpp:99, framefilter.cpp:221-223) */
    p.extendRowBorder = (extendCURowBorder_t)t.extendRowBorder;
}
}

/* the ABI's partition enum must be the reference's */
typedef char x265_hip_check_parts[(int)x265::NUM_LUMA_PARTITIONS == (int)X265_HIP_NUM_LUMA_PARTITIONS ? 1 : -1];
typedef char x265_hip_check_order[(int)x265::LUMA_64x64 == (int)X265_HIP_LUMA_64x64 &&
                                  (int)x265::LUMA_12x16 == (int)X265_HIP_LUMA_12x16 &&
                                  (int)x265::LUMA_16x64 == (int)X265_HIP_LUMA_16x64 ? 1 : -1];
typedef char x265_hip_check_pixel[sizeof(pixel) == 1 ? 1 : -1]; /* 8-bit builds only */
