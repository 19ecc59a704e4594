/*
 * rebvio_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A dependency-free restatement of the reference's per-frame edge-detection + edge-tracking
 * hot path (baumlin/rebvio: scale_space.cpp, edge_detector.cpp, edge_map.cpp, core.hpp
 * DistanceField, core.cpp:33-262,417-456, and the call order of rebvio.cpp:119-292).
 *
 * PARITY UNPINNED: the reference holds no golden vector / KAT for this path (its only
 * fixture is an absent git-LFS bag) and cannot be compiled here (TooN,
 * OpenCV, spdlog absent). The oracle is therefore pinned only by (i) the analytic cases in
 * tests/test_oracle_analytic.py and (ii) the one reference KAT that exists
 * (rebvio/test/test_rebvio.cpp:8-17, estimateLs4Acceleration — host glue, not hot path).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef REBVIO_ORACLE_H_
#define REBVIO_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Field-for-field the reference's types::KeyLine (types/keyline.hpp:24-40), 84 bytes. */
typedef struct orc_keyline {
  float pos[2];
  float pos_img[2];
  float match_pos_img[2];
  float gradient[2];
  float match_gradient[2];
  float gradient_norm;
  float match_gradient_norm;
  float rho;
  float sigma_rho;
  int id;
  int id_prev;
  int id_next;
  int match_id;
  int match_id_forward;
  int match_id_keyframe;
  unsigned int matches;
} orc_keyline;

/* Camera constants (camera.hpp:25-45) + EdgeDetectorConfig (edge_detector.hpp:19-32) +
 * CoreConfig (core.hpp:82-95) + EdgeMapConfig (edge_map.hpp:19-26) + the two ImuStateConfig
 * noise terms used by the visual glue (imu.hpp:158-159). */
typedef struct orc_params {
  int rows, cols;
  float fm, cx, cy;
  int keylines_ref, keylines_max;
  float pos_neg_threshold, dog_threshold, threshold, gain, max_threshold, min_threshold;
  float search_range, reweight_distance, match_treshold;
  unsigned int min_match_threshold, iterations, global_min_matches_threshold;
  float pixel_uncertainty, quantile_cutoff;
  int quantile_num_bins;
  float reshape_q_abs;
  float pixel_uncertainty_match, match_threshold_norm, match_threshold_angle, regularization_threshold;
  float gyro_std_dev, gyro_bias_std_dev;
} orc_params;

typedef struct orc_ctx orc_ctx;
typedef struct orc_map orc_map;

/* Result of one frame-pair tracking step (rebvio.cpp:142-259, visual + gyro-prior part). */
typedef struct orc_pair_out {
  float Vg[3];        /* minimizeVel translation                         */
  float P_Vg[9];      /* minimizeVel covariance (= invert(JtJ))          */
  float F;            /* minimizeVel score                               */
  float Xv[6];        /* extRotVel solution                              */
  float W_Xv[36];     /* extRotVel information matrix (= JtJ)            */
  float Xgv[6];       /* after gyroBiasCorrection                        */
  float V[3];         /* final translation used by directedMatch / EKF   */
  float R[9];         /* corrected inter-frame rotation (= Rgva)         */
  float P_V[9];
  float sigma_rho_min;
  int ext_ok;
  int klm_num;        /* directedMatch count                             */
  int kf_matches;
  int reg_num;        /* regularize1Iter count                           */
  int lm_accept_mask; /* bit i = LM iteration i accepted                 */
  int status;         /* 0 ok, 1 minimization NaN, 2 insufficient matches */
} orc_pair_out;

void orc_default_params(orc_params* p, int rows, int cols);
orc_ctx* orc_create(const orc_params* p);
/* DIAGNOSTICS (not the reference) for the keyline sums of tryVel / extRotVel: 1 = accumulated in double, 2 = the fp32 terms added
   in the HIP kernels' order (rebvio_oracle.cpp, struct Acc); default 0 = fp32 in index order, as the reference adds. */
void orc_set_wide_sums(orc_ctx* c, int on);
void orc_destroy(orc_ctx* c);

/* scale space on an fp32 image (values 0..765), outputs are rows*cols floats (any may be NULL) */
void orc_scale_space(orc_ctx* c, const float* img, float* scale0, float* scale1, float* dog, float* mag);
/* one FastGaussian pass pair exposed for analytic tests: integral image and box average */
void orc_integral_image(int rows, int cols, const float* in, float* out);
void orc_box_average(int rows, int cols, int d, const float* integral, float* out);
int orc_filter_width(orc_ctx* c, int filter, int pass);

/* EdgeDetector::detect: returns a new map owned by the caller (orc_map_free) */
orc_map* orc_detect(orc_ctx* c, const float* img, uint64_t ts_us);
float orc_detector_threshold(orc_ctx* c);      /* config_->threshold after the servo      */
float orc_detector_auto_threshold(orc_ctx* c); /* auto_threshold_                          */
void orc_detector_mask(orc_ctx* c, int* mask_out);

int orc_map_size(orc_map* m);
float orc_map_threshold(orc_map* m);
void orc_map_set_threshold(orc_map* m, float t);
void orc_map_get_keylines(orc_map* m, orc_keyline* out);
void orc_map_set_keylines(orc_map* m, const orc_keyline* in, int n);
void orc_map_get_mask(orc_map* m, int* mask_out); /* dense rows*cols, -1 = none */
orc_map* orc_map_clone(orc_map* m);
void orc_map_free(orc_map* m);

void orc_build_distance_field(orc_ctx* c, orc_map* m);
void orc_distance_field(orc_ctx* c, int* id_out, int* dist_out);

void orc_rotate_keylines(orc_ctx* c, orc_map* m, const float R[9]);
float orc_estimate_quantile(orc_map* m, float percentile, int num_bins);
float orc_try_vel(orc_ctx* c, orc_map* m, const float vel[3], float sigma_rho_min, float* residuals,
                  float JtJ[9], float JtF[3]);
float orc_minimize_vel(orc_ctx* c, orc_map* m, float vel[3], float Rvel[9], int* accept_mask,
                       float* sigma_rho_min_out);
int orc_forward_match(orc_map* old_map, orc_map* new_map);
int orc_ext_rot_vel(orc_ctx* c, const float vel[3], float Wx[36], float X[6], float JtF_out[6]);
int orc_directed_match(orc_ctx* c, orc_map* new_map, orc_map* old_map, const float vel[3],
                       const float Rvel[9], const float Rback[9], int* kf_matches, float max_radius);
int orc_regularize(orc_map* m);
void orc_update_inverse_depth(orc_ctx* c, const float vel[3]);

/* Single-keyline forms of the reference's public methods: EdgeMap::searchMatch (edge_map.cpp:101-184) on `searched`,
 * Core::testfk (core.cpp:39-44), Core::calculatefJ (core.cpp:46-76, on the field of orc_build_distance_field),
 * Core::updateInverseDepthARLU (core.cpp:424-456), FastGaussian(camera, sigma, n).smooth (scale_space.cpp:14-41,173-182). */
int orc_search_match(orc_ctx* c, orc_map* searched, const orc_keyline* query, const float vel[3], const float Rvel[9],
                     const float Rback[9], float max_radius);
int orc_test_fk(const orc_keyline* k1, const orc_keyline* k2, float similarity_threshold);
float orc_calculate_fj(orc_ctx* c, int f_inx, float* df_dx, float* df_dy, orc_keyline* keyline, float px, float py, int* mnum,
                       float* fi);
void orc_update_inverse_depth_arlu(orc_ctx* c, orc_keyline* keyline, const float vel[3]);
void orc_smooth(orc_ctx* c, const float* img, float sigma, int n, float* out, int widths_out[3]);
void orc_smooth_n(orc_ctx* c, const float* img, float sigma, int n, float* out, int* widths_out /* [n] */);

/* Persistent gyro-bias state of the glue (imu.hpp:180-183): reset to the reference's initial values */
void orc_reset_state(orc_ctx* c);
/* One frame pair: rebvio.cpp:142-259 without the accelerometer/SAB fusion (the branch taken for the
 * first 4+init_bias_frame_num frames). R_prior = IMU inter-frame rotation (NULL = identity). */
int orc_track_pair(orc_ctx* c, orc_map* old_map, orc_map* new_map, const float* R_prior, float frame_dt,
                   orc_pair_out* out);

/* ---- full VIO glue (SURVEY.md N2 / BASELINE config 5): rebvio.cpp:92-293 with IMU pre-integration (types/imu.hpp),
 * gyro-bias initialisation, Core::estimateBias + SABEstimator (core.cpp:350-414, sab_estimator.cpp) and pose integration.
 * IMU samples: ts[us], gyro[3], acc[3] in the IMU frame; R_c2i / t_c2i = camera->IMU extrinsics (camera.hpp:41-45). */
typedef struct orc_vio_out {
  orc_pair_out pair;
  float orientation[3]; /* so3 log of R_global (odometry.orientation) */
  float position[3];
  float K;              /* scale tan(X[0]) */
  float g_est[3], b_est[3], Bg[3];
  int initialized;      /* imu_state_.initialized */
  int sab_active;       /* estimateBias branch taken this frame */
} orc_vio_out;
void orc_vio_reset(orc_ctx* c, const float R_c2i[9], const float t_c2i[3]);
/* attach the IMU samples with ts <= the map's frame timestamp to `m` (rebvio.cpp:77-84) */
void orc_vio_add_imu(orc_ctx* c, orc_map* m, uint64_t ts_us, const float gyro[3], const float acc[3]);
int orc_vio_step(orc_ctx* c, orc_map* old_map, orc_map* new_map, orc_vio_out* out);

/* Front end (rebvio.cpp:43-47): convertTo(CV_32F,3.0) + cv::undistort with K4 = fx,fy,cx,cy and
 * D5 = k1,k2,p1,p2,k3 (camera.hpp:39-40). out = rows*cols floats. */
void orc_front_end_u8(orc_ctx* c, const uint8_t* img, const float K4[4], const float D5[5], float* out);

/* host glue pieces exposed for KATs */
void orc_ls4_reset(orc_ctx* c);
void orc_estimate_ls4_acceleration(orc_ctx* c, const float vel[3], float acc[3], const float R[9], float dt);
void orc_so3_exp(const float w[3], float R[9]);
void orc_sym6_solve(const float A[36], const float b[6], float x[6]);

/* CPU-baseline driver: runs nframes u8 frames (frame i at frames + idx[i]*rows*cols) through
 * detect + track_pair. threads = 1 (serial) or 2 (detect || track, like rebvio.cpp:28-29).
 * Returns wall seconds; optional per-frame keyline counts / match counts. */
double orc_run_stream(orc_ctx* c, const uint8_t* frames, const int* idx, int nframes, int threads,
                      int* keyline_counts, int* match_counts, float* pose_out /* nframes*6: Vg,dWgv */);
/* The same, additionally reporting when each frame was finished (seconds since the start; frame time percentiles). */
double orc_run_stream_ex(orc_ctx* c, const uint8_t* frames, const int* idx, int nframes, int threads,
                         int* keyline_counts, int* match_counts, float* pose_out, double* frame_done_s);
/* Wall seconds accumulated so far at the reference's REBVIO_TIMER tick sites (util/timer.hpp:18-32): out[0] detect
 * (edge_detector.cpp:31,41), [1] buildDistanceField (core.cpp:34-36), [2] minimizeVel (core.cpp:152,187), [3] extRotVel
 * (core.cpp:193,258), [4] directedMatch (edge_map.cpp:189,216), [5] the rest of the pair step. reset != 0 clears them. */
void orc_stage_seconds(orc_ctx* c, double out[6], int reset);

#ifdef __cplusplus
}
#endif
#endif
