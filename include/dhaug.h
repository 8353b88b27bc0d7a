/*
 * dhaug.h -- C-ABI of libdhaug.so: hand-written HIP (gfx950 / MI355X) kernels for the DH-AUG hot path
 * (DH forward kinematics + MLP GAN generator / critic step).
 *
 * The reference (R/ = DH-AUG_master/) is pure Python on stock PyTorch ops; it has no FFI layer.  Each entry
 * point below therefore names the reference *Python* symbol whose arithmetic it replaces (file:line).  The
 * Python drop-in classes in the package bind these through ctypes.
 *
 * Conventions
 *   - plain C: raw device pointers + sizes; no torch / C++ types cross the boundary.
 *   - the caller owns every buffer; the library allocates nothing and keeps no state.
 *   - all work is enqueued on the caller's HIP stream (`stream`, a hipStream_t passed as void*); no
 *     synchronisation inside, safe for hipGraph capture.
 *   - return value: 0 on success; a negative DHAUG_E* code for an argument error detected on the host
 *     (nothing was launched); a positive hipError_t value if the launch failed.  Never throws.
 *   - fp32 tensors are contiguous row-major.  bf16 tensors are row-major with an explicit leading dimension
 *     (elements); rows must start 16-byte aligned (ld % 8 == 0, base 16-byte aligned).
 *   - angle layout (degrees), 37 columns = `generator_angle` of R/models_Fk_GAN/Fk_generator.py:179-186:
 *       [0:5] right leg, [5:10] left leg, [10:23] body, [23:28] right arm, [28:33] left arm,
 *       [33] unused, [34:37] global rotation x,y,z.
 *   - bone_len layout, 15 columns = used_16key_15bone_len_table order
 *       (R/models_Fk_GAN/forward_kinematics_DH_model.py:46-49).
 *   - out16 = the 16 H36M joints of R/common/h36m_dataset.py:37-38, (N,16,3).
 */
#ifndef DHAUG_H
#define DHAUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHAUG_VERSION 100            /* major*10000 + minor*100 + patch */

#define DHAUG_OK             0
#define DHAUG_EINVAL        -1       /* null pointer / negative size / bad enum */
#define DHAUG_EALIGN        -2       /* pointer or leading dimension violates the alignment contract */
#define DHAUG_EUNSUPPORTED  -3       /* shape outside what the kernels implement */

/* activation kinds for dhaug_gemm_bf16 epilogues */
#define DHAUG_ACT_NONE   0
#define DHAUG_ACT_RELU   1
#define DHAUG_ACT_LRELU  2           /* LeakyReLU, slope passed separately (reference uses 0.01) */

int dhaug_version(void);
/* name of the code-object architecture the library was built for ("gfx950") */
const char* dhaug_arch(void);

/* ------------------------------------------------------------------------------------------------------
 * Forward kinematics
 * ---------------------------------------------------------------------------------------------------- */

/* Fused DH forward kinematics: 33 local DH matrices, 5 chain products, global rotation, root add,
 * 32->16 joint gather -- one kernel, one HBM read of the inputs, one write of the joints.
 * Replaces Forward_Kinematics_DH_Model.change_3d_joint_angle (torch branch)
 *   R/models_Fk_GAN/forward_kinematics_DH_model.py:562-822 (+ dh_matrix :80-116, rotationMatrix :141-191)
 * followed by the H36M_32_To_16_Table gather of R/models_Fk_GAN/Fk_generator.py:259.
 *   angles (N,37) deg, bone_len (N,15) m, root (N,3) m  ->  out (N,out_joints,3), out_joints in {16,32}.
 * out_joints == 32 reproduces the reference's (N,32,3) tensor exactly (unused rows = root). */
int dhaug_fk_forward(const float* angles, const float* bone_len, const float* root, float* out,
                     int64_t N, int out_joints, void* stream);

/* Reverse-mode gradient of dhaug_fk_forward(out_joints=16) w.r.t. all three inputs (the G step
 * back-propagates through FK: R/models_Fk_GAN/model_fk_gan_train.py:431-480).
 *   grad_out16 (N,16,3) -> grad_angles (N,37), grad_bone_len (N,15), grad_root (N,3).
 * grad_angles columns that cannot move any joint (4,9,22,27,32,33) are written as 0. */
int dhaug_fk_backward(const float* angles, const float* bone_len, const float* grad_out16,
                      float* grad_angles, float* grad_bone_len, float* grad_root, int64_t N, void* stream);

/* Generator tail fused in front of FK: tanh on the 35 head columns (x10 on the root columns), scatter of the
 * 31 live columns into the 37 angle slots, joint-limit affine map (use_preangle) or x180, bone-length
 * jitter len*(1+s[pair]), FK, 32->16 gather.
 * Replaces R/models_Fk_GAN/Fk_generator.py:121-259 (Fk_Generator.forward after deconv_out) and :310-453.
 *   head (N,35) fp32 pre-activation, bone_len (N,15), scaler (N,8) (may be NULL = no jitter)
 *   -> fake16 (N,16,3); angles_out (N,37) optional (NULL to skip) = `generator_angle`. */
int dhaug_gen_tail_forward(const float* head, const float* bone_len, const float* scaler, float* fake16,
                           float* angles_out, int64_t N, int use_preangle, void* stream);

/* dhaug_gen_tail_forward that also emits what the critics consume, while the joints are still in registers:
 *   centered (N,16,3) = fake16 - fake16[:,0]                      (R/models_Fk_GAN/model_fk_gan_train.py:312)
 *   kcs_bf16 (N,32) bf16 = special_KCS_Input_transform(fake16) zero-padded (Fk_discriminator.py:36-146; the operand of
 *                          dhaug_kcs_forward / dhaug_mlp_forward)
 *   proj2d   (N,16,2)   = project_to_2d(GAN_torch_world_to_camera(fake16, quat, trans), cam9)   (:374-376;
 *                          quat[4], trans[3], cam9[9] host arrays as in dhaug_world_to_camera_project)
 * Any of the three may be NULL.  draw_scaler != 0 (scaler must be NULL): the bone-length jitter
 * `randint(-200, 200, (N, 8)) / 1000` (Fk_generator.py:196-203) is drawn inside the kernel with Philox4x32-10 keyed by
 * rng_seed, counter (pose index, rng_offset); scaler_out (N,8), if given, receives the draw.
 * inputs_bf16 != 0: centered and proj2d are written as bf16 (N,48) / (N,32) -- the rounding the critics' first layer
 * applies to them anyway (dhaug_mlp_forward LOAD units), at half the bytes. */
int dhaug_gen_tail_forward_critics(const float* head, const float* bone_len, const float* scaler, float* fake16,
                                   void* centered, uint16_t* kcs_bf16, const float* quat, const float* trans,
                                   const float* cam9, void* proj2d, int draw_scaler, uint64_t rng_seed,
                                   uint64_t rng_offset, float* scaler_out, int64_t N, int use_preangle, int inputs_bf16,
                                   void* stream);

/* Gradient of dhaug_gen_tail_forward w.r.t. head: grad_fake16 (N,16,3) -> grad_head (N,35)
 * (column 31 = 0).  Recomputes the forward from head. */
int dhaug_gen_tail_backward(const float* head, const float* bone_len, const float* scaler,
                            const float* grad_fake16, float* grad_head, int64_t N, int use_preangle,
                            void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Pose features
 * ---------------------------------------------------------------------------------------------------- */

/* Bone lengths of 16-joint poses: (N,16,3) -> (N,15).  Replaces
 * Fk_Generator.GAN_generator_get_bone_length, R/models_Fk_GAN/Fk_generator.py:107-111
 * (Fk_get_boneVecByPose3d, R/models_Fk_GAN/special_operate.py:513-539, + norm). */
int dhaug_bone_length(const float* pose16, float* bone_len, int64_t N, void* stream);

/* KCS features: 15 adjacent-bone cosines (+ the 15 bone lengths when with_lengths != 0).
 * Replaces special_KCS_Input_transform R/models_Fk_GAN/Fk_discriminator.py:36-146 and the video variant
 * :269-377.  pose16 (N,16,3) fp32.
 *   out_f32 : optional (N, 30|15) fp32, contiguous.
 *   out_bf16: optional (N, ld_bf16) bf16 with columns [30|15, ld_bf16) zero-filled (GEMM operand). */
int dhaug_kcs_forward(const float* pose16, float* out_f32, uint16_t* out_bf16, int64_t ld_bf16,
                      int64_t N, int with_lengths, void* stream);

/* The 3D critic's two inputs in one pass over the pose: centered (N,16,3) = pose16 - pose16[:,0] (the
 * `inputs_3d - inputs_3d[:, :1]` at R/models_Fk_GAN/model_fk_gan_train.py critic calls) and the bf16 KCS operand of
 * dhaug_kcs_forward (KCS is translation invariant, so it is the same feature the critic computes from `centered`). */
int dhaug_center_kcs_forward(const float* pose16, float* centered, uint16_t* out_bf16, int64_t ld_bf16,
                             int64_t N, int with_lengths, void* stream);

/* VJP of dhaug_kcs_forward: grad_feat (N,30|15) fp32 -> grad_pose16 (N,16,3). */
int dhaug_kcs_backward(const float* pose16, const float* grad_feat, float* grad_pose16, int64_t N,
                       int with_lengths, void* stream);

/* JVP (forward-mode) of dhaug_kcs_forward along tangent (N,16,3): -> tan_feat (N,30|15).  Used by the
 * analytic WGAN-GP gradient (R/models_Fk_GAN/Fk_discriminator.py:205-231 with create_graph=True). */
int dhaug_kcs_jvp(const float* pose16, const float* tangent, float* tan_feat, int64_t N, int with_lengths,
                  void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Camera / projection ("next" row N1): between the FK output and the 2D critic on every iteration
 * ---------------------------------------------------------------------------------------------------- */

/* world -> camera by quaternion (qinverse, qrot) + H36M non-linear projection, one shared camera.
 * Replaces GAN_torch_world_to_camera R/common/camera.py:36-38 (+ R/common/quaternion.py:6-35) and
 * project_to_2d R/common/camera.py:62-94 as called at R/models_Fk_GAN/model_fk_gan_train.py:374-376.
 *   pose16 (N,16,3) world; quat[4] (w,x,y,z), trans[3], cam9[9] = f(2) c(2) k(3) p(2)  (host arrays)
 *   -> cam3d (N,16,3) optional, proj2d (N,16,2) optional. */
int dhaug_world_to_camera_project(const float* pose16, const float* quat, const float* trans,
                                  const float* cam9, float* cam3d, float* proj2d, int64_t N, void* stream);

/* VJP of the above w.r.t. pose16: grad_cam3d / grad_proj2d (either may be NULL) -> grad_pose16. */
int dhaug_world_to_camera_project_backward(const float* pose16, const float* quat, const float* trans,
                                           const float* cam9, const float* grad_cam3d,
                                           const float* grad_proj2d, float* grad_pose16, int64_t N,
                                           void* stream);

/* camera -> world with per-sample quaternion/translation (N,4),(N,3):
 * GAN_torch_camera_to_world_batch R/common/camera.py:53-59. */
int dhaug_camera_to_world(const float* cam3d, const float* quat, const float* trans, float* world,
                          int64_t N, void* stream);

/* Real-data side of the augmentation API ("next" row N3).
 * dhaug_bone_length_swap: random_bl_aug of R/function_aug/dataloader_update.py:18-40 -- every bone keeps its direction
 * and takes the length new_len[k] (N,15 in PoseAug bone order: (0,1)(1,2)(2,3)(0,4)(4,5)(5,6)(0,7)(7,8)(8,9)(8,10)
 * (10,11)(11,12)(8,13)(13,14)(14,15), R/utils/gan_utils.py:90-138); the pose is rebuilt from the root down.
 * dhaug_project_to_2d: project_to_2d (R/common/camera.py:62-94) of camera-space poses with per-sample intrinsics
 * cam9 (N,9) on the device, as called at dataloader_update.py:69. */
int dhaug_bone_length_swap(const float* pose16, const float* new_len, float* out, int64_t N, void* stream);
int dhaug_project_to_2d(const float* cam3d, const float* cam9, float* proj2d, int64_t N, void* stream);

/* Root-centre and/or left/right flip of (N,16,C) poses, C in {2,3}:
 * x - x[:, :1] (R/models_Fk_GAN/model_fk_gan_train.py:295,312) and the flip of :320-331
 * (negate coordinate 0, swap joints [4,5,6,10,11,12] <-> [1,2,3,13,14,15]). */
int dhaug_center_flip(const float* in, float* out, int64_t N, int C, int center, int flip, void* stream);
/* transpose (VJP) of the linear map above */
int dhaug_center_flip_backward(const float* grad_out, float* grad_in, int64_t N, int C, int center, int flip,
                               void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Dense layers: bf16 MFMA GEMM with fused epilogue
 * ---------------------------------------------------------------------------------------------------- */

/* C[M,N] = act( A[M,K] * B[N,K]^T + bias[N] + residual[M,N] )      (v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16, fp32 acc)
 * A, B bf16 row-major, contraction dimension contiguous in both (nn.Linear weight layout [out,in]).
 * Replaces nn.Linear + ReLU/LeakyReLU + residual add of myResNet (R/models_Fk_GAN/special_operate.py:490-510)
 * and the Linear stacks of R/models_Fk_GAN/Fk_generator.py:95-103, Fk_discriminator.py:156-178,243-249.
 *   K % 16 == 0 (operands zero-padded by their producers); lda, ldb, ld_res, ldc_* in elements, % 8 == 0.
 *   bias     : fp32 [N] or NULL.
 *   residual : bf16 (M, ld_res) or NULL, added before the activation.
 *   residual_f32 : fp32 (M, ld_res_f32) or NULL, same role (fp32-grade "bf16x3" path).
 *   c_bf16   : optional bf16 output (M, ldc_bf16); columns [N, n_pad_zero) are written as 0 so the result
 *              can feed the next GEMM as a zero-padded operand.
 *   c_f32    : optional fp32 output (M, ldc_f32).
 * At least one output must be given.
 * READ EXTENT (every dhaug_gemm_bf16* entry point, dhaug_gemm_bf16_group members included): each of the M rows of A and of the N
 * rows of B is read over EXACTLY K columns -- elements [r * ld, r * ld + K) of row r, nothing beyond them (no kernel reads past
 * column K; the K tail of a tile is clamped or zero-sourced) -- and the rows of residual / dmask over N columns.  A caller that
 * passes a K wider than the data of a row (a 1000-column block of a wider buffer contracted with K = 1008 against zero weight
 * columns) must therefore own all K columns of the LAST row too, and they must hold finite values: 0 * NaN = NaN.  The Python
 * host layer gives such buffers a guard row (critic_step._Math.empty_blocks, gen_step); tests/test_gpu_graphs.py::
 * test_whole_iterations_read_no_unwritten_memory and tests/test_gpu_gemm_p8.py::test_p8_reads_nothing_beyond_k hold both sides.
 * Kernels: 256-wide layers of whole 32-row tiles -> gemm_nt256s; K <= 256 -> the weight-stationary kernel; wide layers with tiles
 * enough for the card (and grouped members of >= 1024 rows) -> the 256 x 256 x 64 ping-pong kernel (csrc/dhaug_gemm_p8.hip);
 * otherwise 64 x 64 / 128 x 128 pipelined tiles.  All sum k ascending into one fp32 accumulator per output. */
int dhaug_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                    const float* bias, const uint16_t* residual, int64_t ld_res,
                    const float* residual_f32, int64_t ld_res_f32,
                    uint16_t* c_bf16, int64_t ldc_bf16, int64_t n_pad_zero,
                    float* c_f32, int64_t ldc_f32,
                    int64_t M, int64_t N, int64_t K, int act, float slope, void* stream);

/* The split-operand ("bf16x3" / "bf16x6") form of the same step: fp32 result, fp32 residual, and the mask read from the producing
 * layer's fp32 activation: c_f32 = (A B^T + residual_f32) * dmask_act'(dmask), dmask (M, >= N) fp32 (ReLU / LeakyReLU only).  A, B:
 * the operands dhaug_split_bf16 makes (K = terms * padded width).  One launch instead of the GEMM + dhaug_act_backward_f32. */
int dhaug_gemm_bf16_dmask_f32(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* residual_f32, int64_t ld_res_f32,
                              const float* dmask, int64_t ld_dmask, int dmask_act, float dmask_slope, float* c_f32, int64_t ldc_f32,
                              int64_t M, int64_t N, int64_t K, void* stream);

/* Input gradient of a layer whose input came out of an activation, in one pass:
 *   c_bf16[M,N] = ( A[M,K] * B[N,K]^T + residual[M,N] ) * act'(dmask[M,N])      act'(y) = y > 0 ? 1 : (ReLU 0 | LeakyReLU slope)
 * i.e. LinearT followed by the activation backward of the producing layer (the `loss.backward()` of
 * R/models_Fk_GAN/special_operate.py:490-510 myResNet: gh * relu'(h)); dmask_act = DHAUG_ACT_NONE gives the plain
 * product.  256-wide layers apply the mask in the GEMM epilogue, other shapes run the GEMM and
 * dhaug_act_backward_bf16 in place. */
int dhaug_gemm_bf16_dmask(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* residual,
                          int64_t ld_res, const uint16_t* dmask, int64_t ld_dmask, int dmask_act, float dmask_slope,
                          uint16_t* c_bf16, int64_t ldc_bf16, int64_t M, int64_t N, int64_t K, void* stream);

/* Same with columns [N, n_pad_zero) of c_bf16 written as 0 (the result feeds the next GEMM as a zero-padded operand) and
 * any N: every kernel applies the mask in its epilogue. */
int dhaug_gemm_bf16_dmask_pad(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* residual,
                              int64_t ld_res, const uint16_t* dmask, int64_t ld_dmask, int dmask_act, float dmask_slope,
                              uint16_t* c_bf16, int64_t ldc_bf16, int64_t n_pad_zero, int64_t M, int64_t N, int64_t K,
                              void* stream);

/* The 256 -> 256 input-gradient / tangent step with the activation mask as a SIGN-BIT array:
 *   c = (A * B^T + residual) * act'(y),   act'(y) read as one bit per element, (y > 0), from `bits` -- the array a
 * forward-with-save layer leaves beside its image (struct dhaug_mlp_unit.bits: same layout, same 32-row tiles; pass the
 * address of the word of the first row's tile when A starts at a later row, a multiple of 32).  M a multiple of 32,
 * N = K = 256, bf16 in / out.  Same arithmetic as dhaug_gemm_bf16_dmask; the mask costs 32 bytes per row instead of 512. */
int dhaug_gemm_bf16_dbits(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* residual, int64_t ld_res,
                          const uint32_t* bits, int dmask_act, float dmask_slope, uint16_t* c_bf16, int64_t ldc_bf16, int64_t M,
                          void* stream);

/* (A B^T) * act'(mask) for an output of one or two 256-wide column blocks whose masks are sign-bit arrays (bits_lo: columns
 * 0..255, bits_hi: 256..511; N = 256 or 512, K <= 256 one of 16, 32, 48, 64, 112, 128, 256): the input-gradient step through
 * the 3D critic's merge layer, R/models_Fk_GAN/Fk_discriminator.py:192-197 under loss.backward() -- the cotangent of the two
 * concatenated branch outputs, each masked by ITS branch's last ReLU. */
int dhaug_gemm_bf16_dbits_wide(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint32_t* bits_lo,
                               const uint32_t* bits_hi, int dmask_act, float dmask_slope, uint16_t* c_bf16, int64_t ldc_bf16,
                               int64_t M, int64_t N, int64_t K, void* stream);

/* The one-workgroup-per-CU (persistent) launches of this library -- the fused programs, the 256-wide layer kernels, the block
 * kernel, the grouped weight gradients -- take at most n workgroups from now on (0 or 256: the whole card); returns the
 * previous value.  Process-wide, read at launch time: set it around the launches of one stream and to the complement around
 * another stream's, and the two chains run side by side on disjoint sets of CUs instead of queueing for the whole card. */
int dhaug_set_workgroup_cap(int n);

/* Non-finite values in the fused INFERENCE programs (dhaug_mlp_forward without save targets, dhaug_mlp_forward_x3).  Default
 * (0): their ReLU is an integer max on the bit pattern -- one instruction in the matrix pipe's shadow -- which lets +NaN pass
 * and turns the matrix pipe's -NaN into 0: a NaN / inf input row or a NaN weight does NOT reach the logit through a ReLU
 * layer (LeakyReLU layers, i.e. the 2D critic, always propagate).  on != 0: every ReLU is applied as max(v, v * 0) in fp32
 * like the forward-with-save unit of the training steps does: a NaN / inf input row gives NaN in ITS logit (no other), NaN
 * weights give NaN logits everywhere -- what R/models_Fk_GAN/Fk_discriminator.py:180-201 does through ATen (inf * 0 = NaN:
 * an inf activation becomes NaN one layer earlier than in the reference, the logit is NaN either way).  Same results for
 * finite values; measured cost on MI355X: bf16 programs +5 % (D3 114 -> 120 us at B = 65 536).  Process-wide, read at launch
 * time; returns the previous value.  The Python package sets it from DHAUG_NAN_PROPAGATION=1 at load. */
int dhaug_set_nan_propagation(int on);

/* Two 256 -> 256 layers of a residual block in one launch (the backward step through myResNet,
 * R/models_Fk_GAN/special_operate.py:490-510 under loss.backward(), and its tangent twin in the gradient penalty's
 * double backward, R/models_Fk_GAN/Fk_discriminator.py:205-231):
 *     Y1 = (X W1^T) * act'(bits1)        Y2 = (Y1 W2^T + X) * act'(bits2)
 * X, Y1, Y2: bf16 (M, ld >= 256), M a multiple of 32, three distinct buffers; W1, W2: bf16 (256, ld >= 256) with the
 * contraction along the row (the operand dhaug_gemm_bf16_dbits takes as B); bits1, bits2: sign-bit arrays of the layers'
 * saved activations (struct dhaug_mlp_unit.bits).  Y1 never leaves LDS between the layers and X is read once: 300 MB
 * per 3 x 65 536-row block instead of the 500 MB of two dhaug_gemm_bf16_dbits launches. */
int dhaug_gemm_block2_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W1, int64_t ldw1, const uint16_t* W2, int64_t ldw2,
                           const uint32_t* bits1, const uint32_t* bits2, int mask_act, float mask_slope,
                           uint16_t* Y1, int64_t ldy1, uint16_t* Y2, int64_t ldy2, int64_t M, void* stream);
/* A chain of such blocks in one launch: block i + 1 takes block i's Y2 as its X (the three myResNet blocks of a branch of the
 * critics, backward: block 3, 2, 1; tangent: 1, 2, 3).  Row tiles are independent and a workgroup keeps its tiles through the
 * chain, so the launch needs no device-wide synchronisation between the blocks.  `blocks`: host array, read during the call. */
#define DHAUG_BLOCK2_MAX 3
typedef struct dhaug_block2 {
    const uint16_t* W1; int64_t ldw1;
    const uint16_t* W2; int64_t ldw2;
    const uint32_t* bits1; const uint32_t* bits2;
    uint16_t* Y1; int64_t ldy1;
    uint16_t* Y2; int64_t ldy2;
} dhaug_block2;
int dhaug_gemm_block2_stack_bf16(const uint16_t* X, int64_t ldx, const dhaug_block2* blocks, int nb, int mask_act, float mask_slope,
                                 int64_t M, void* stream);

/* Weight-gradient GEMM: C[N1,N2] (+)= A[M,N1]^T * B[M,N2], contraction over the batch dimension M
 * (split across workgroups, fp32 atomics into C).  bf16 operands, fp32 result.  `accumulate` != 0 adds
 * into C, otherwise C is overwritten (zeroed by the library on the stream first).  colsum_a (optional, fp32 [N1])
 * receives the column sums of A over M -- the bias gradient of the same layer -- computed on the matrix pipe as
 * A^T * ones by the workgroups that already hold the A tiles (same accumulate / overwrite rule as C). */
int dhaug_gemm_tn_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                       float* C, int64_t ldc, float* colsum_a, int64_t M, int64_t N1, int64_t N2, int accumulate,
                       void* stream);
/* Same, with the column sums restricted to rows [0, colsum_rows) of A (colsum_rows == M or a multiple of 128): the
 * explicit critic step contracts real, fake AND interpolated rows in one launch, but only the real / fake rows carry a
 * bias gradient (the gradient penalty has none). */
int dhaug_gemm_tn_bf16_rows(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                            float* C, int64_t ldc, float* colsum_a, int64_t colsum_rows, int64_t M, int64_t N1, int64_t N2,
                            int accumulate, void* stream);

/* n <= 8 independent GEMMs of the same (M, N, K) as ONE launch (64 x 64 tiles, blockIdx.y = member): member i computes what
 * dhaug_gemm_bf16 (dmask == NULL or dmask_act == DHAUG_ACT_NONE) / dhaug_gemm_bf16_dmask_pad compute from the same fields --
 * c = act(A B^T + bias + residual + residual_f32), then * dmask_act'(dmask).  For the branches of a motion critic
 * (R/models_Fk_GAN/Fk_discriminator.py:381-587: four / two stacks of identical layers over different features): a 1 536 x 1000 x 1000
 * layer alone is one and a half waves of the card's workgroup slots behind a launch of its own.  N > 64, K >= 64; everything else
 * as dhaug_gemm_bf16. */
typedef struct dhaug_gemm_desc {
    const uint16_t* A; int64_t lda;
    const uint16_t* B; int64_t ldb;
    const float* bias;
    const uint16_t* residual; int64_t ld_res;
    const float* residual_f32; int64_t ld_res_f32;
    uint16_t* c_bf16; int64_t ldc_bf16; int64_t n_pad_zero;
    float* c_f32; int64_t ldc_f32;
    int64_t M, N, K;
    int32_t act; float slope;
    const uint16_t* dmask; int64_t ld_dmask; int32_t dmask_act; float dmask_slope;
} dhaug_gemm_desc;
int dhaug_gemm_bf16_group(const dhaug_gemm_desc* members, int n, void* stream);

/* The weight / bias gradients of ALL layers of a critic step in one launch (+ one small launch that sums the partial
 * results): layer i computes C[N1,N2] (+)= A[M,N1]^T * B[M,N2] and colsum_a[N1] (+)= column sums of A over rows
 * [0, colsum_rows), M and colsum_rows multiples of 32, operands bf16 with rows readable up to ceil8(N) columns (of every
 * 256-column block).  N1, N2 <= 256, or "wide" (up to 4096: a grid of 256 x 256 blocks, one workgroup each, phase 0 only -- for
 * groups that have blocks enough to fill the card without splitting any over the batch: the DenseDim-1000 layers of a video step).
 * A block that ends up with ONE workgroup adds its result into C / colsum_a itself (no partial result, no sum).  One workgroup per CU owns a layer's WHOLE output over its slice of the batch (every operand byte crosses
 * L2 -> LDS once; the 64 x 64-tile kernel behind dhaug_gemm_tn_bf16 re-reads each row four times); the workgroups are dealt
 * out to the layers by what their 32-row stages cost (operand bytes, with a floor for narrow layers), so a step leaves 256
 * partial results in total, not per layer.
 * `layers`: host array (read during the call).  `workspace`: DHAUG_TN_GROUP_WORKSPACE_FLOATS fp32 values owned by the
 * caller, any content (calls sharing it must be ordered on one stream).
 * The outputs of ONE call must be distinct: no two layers may name the same C or the same colsum_a (they are summed
 * concurrently, a one-workgroup block with a plain read-modify-write) -- DHAUG_EINVAL otherwise.  A second contribution
 * to a gradient goes into a second call (stream order is the synchronisation).
 * Replaces the parameter-gradient half of loss.backward() in R/models_Fk_GAN/model_fk_gan_train.py:191-214. */
#define DHAUG_TN_GROUP_MAX 42
#define DHAUG_TN_GROUP_WORKSPACE_FLOATS (256LL * (256 * 256 + 256))
typedef struct dhaug_tn_layer {
    const uint16_t* A; int64_t lda;
    const uint16_t* B; int64_t ldb;
    float* C; int64_t ldc;
    float* colsum_a;            /* optional */
    int64_t colsum_rows;
    int64_t M;
    int32_t N1, N2;
    int32_t accumulate;         /* != 0: add into C / colsum_a, else overwrite */
    int32_t max_workgroups;     /* of layers[0]: the launch uses at most this many workgroups (0: one per CU) -- a launch that
                                   runs beside other kernels leaves them CUs */
    int32_t planes_a, planes_b; /* 0: A / B is an ordinary (M, N) operand.  1 / 2: it is dhaug_split_bf16(x, mode 2, terms 6) seen as
                                   (M / 2, ceil16 N) rows -- the three distinct pieces [hi | mid | lo] of every tensor row -- and the
                                   contraction runs over M = 6 x (tensor rows) VIRTUAL rows, virtual row 6 m + t being piece
                                   (0 0 1 1 0 2)[t] (1: the order of the mode 0 split) or (0 1 0 1 2 0)[t] (2: mode 1) of tensor row m:
                                   the same sum, bit for bit, as over the mode 0 / mode 1 operand seen as (6 x rows, ceil16 N), from half
                                   the bytes (the "bf16x6" weight gradients, autograd_ops._raw_outer).  M % 6 == 0 then. */
} dhaug_tn_layer;
int dhaug_gemm_tn_group_bf16(const dhaug_tn_layer* layers, int n, float* workspace, void* stream);
/* The same in two calls: phase 1 launches the contractions (partial results into `workspace`), phase 2 the sums into the
 * gradient slots -- same `layers`, same `workspace`; phase 0 = both.  Only the sums touch C / colsum_a, so the contractions
 * of one part of a step may run beside another part's (R/models_Fk_GAN/model_fk_gan_train.py:191-214: the three backward
 * passes of a critic step add into the same .grad). */
int dhaug_gemm_tn_group_bf16_phase(const dhaug_tn_layer* layers, int n, float* workspace, int phase, void* stream);

/* fp32 -> bf16 (round-to-nearest-even) with zero padding: src (rows, cols) ld_src -> dst (rows, ld_dst),
 * columns [cols, pad_cols) zero-filled.  Used to pack weights / inputs as GEMM operands. */
int dhaug_cast_pad_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst,
                        int64_t rows, int64_t cols, int64_t pad_cols, void* stream);

/* Same with transposition: src (rows, cols) -> dst (cols, ld_dst) = src^T, columns [rows, pad_cols) zero. */
int dhaug_cast_transpose_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst,
                              int64_t rows, int64_t cols, int64_t pad_cols, void* stream);

/* Operand split for fp32-grade products on the bf16 MFMA path (the parity modes "bf16x3" / "bf16x6").
 *   terms 3: x = hi + lo.         mode 0 (activation side) row = [hi|hi|lo],            mode 1 (weight side) [hi|lo|hi]
 *   terms 6: x = hi + mid + lo.   mode 0 row = [hi|hi|mid|mid|hi|lo],                   mode 1 [hi|mid|hi|mid|lo|hi]
 * each segment pad_cols wide (zero padded), so that A' * B'^T sums every product term down to 2^-16 (terms 3) or
 * 2^-24 (terms 6) relative.  dst is (rows, terms*pad_cols) contiguous. */
int dhaug_split_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t rows, int64_t cols,
                     int64_t pad_cols, int mode, int terms, void* stream);
/* (terms 6 only) mode 2: the three DISTINCT pieces once, row = [hi|mid|lo] (dst is (rows, 3*pad_cols)): half the bytes of mode 0.  The
 * operand of dhaug_gemm_bf16x6_planes, which reads piece (0 0 1 1 0 2)[s] for K-segment s -- the mode 0 row without its copies. */

/* c_f32[M,N] = act( X W^T + bias + residual_f32 ) * (dmask_f32 > 0 ? 1 : dneg) in the "bf16x6" arithmetic with the activation side as
 * PLANES: A_planes = dhaug_split_bf16(x, mode 2, terms 6) = [hi|mid|lo] of kp columns each (lda >= 3 kp), B = dhaug_split_bf16(W, mode 1,
 * terms 6) (ldb >= 6 kp) with x_order 0; x_order 1: the planes stand for the mode 1 operand and B is the mode 0 split (the backward
 * chain: the cotangent is split once, in the layout the weight gradients contract it in).
 * The same six product terms in the same order as dhaug_gemm_bf16 on the mode 0 / mode 1 operand (K = 6 kp) -- BIT-identical
 * results -- but the split writes 6 instead of 12 bytes per value and the GEMM's re-reads of a piece come from L2 / MALL, not from HBM.
 * kp = 64 * 2^j (a K-tile never straddles two pieces); runs on the 256 x 256 x 64 ping-pong tiles only (csrc/dhaug_gemm_p8.hip): N % 8 == 0,
 * 16-byte aligned bias / residual / mask / output with row pitches % 4 == 0, DHAUG_EUNSUPPORTED otherwise (the caller then splits with
 * mode 0 and calls dhaug_gemm_bf16).  dmask_act DHAUG_ACT_NONE: no mask.
 * x_order 2: A_planes is the ordinary six-segment operand (mode 0, lda >= 6 kp, any kp % 8 == 0 with 6 kp >= 128: a narrow input
 * layer whose RESULT is wanted as planes).  c_planes (optional; ld_planes >= 3 N, % 8 == 0, 16-byte
 * aligned): the result once more as planes [hi|mid|lo] of N columns each -- bit for bit what dhaug_split_bf16(c_f32, mode 2, terms 6,
 * pad_cols N) makes of it, written by the epilogue that holds the values: the next layer's operand without a split launch.  Replaces, with the split, the fp32 layer products of the
 * training passes, R/models_Fk_GAN/model_fk_gan_train.py:177-230. */
int dhaug_gemm_bf16x6_planes(const uint16_t* A_planes, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                             const float* residual_f32, int64_t ld_res_f32, const float* dmask_f32, int64_t ld_dmask_f32, int dmask_act,
                             float dmask_slope, float* c_f32, int64_t ldc_f32, uint16_t* c_planes, int64_t ld_planes, int64_t M, int64_t N,
                             int64_t kp, int x_order, int act, float slope, void* stream);

/* The same split with IEEE-half pieces, x = hi + lo (22 significant bits; three product terms: mode 0 row = [hi|hi|lo], mode 1
 * [hi|lo|hi]; mode 2: the two distinct pieces once, [hi|lo], dst (rows, 2*pad_cols)): the operands of dhaug_gemm_f16x3[_planes].
 * |x| < 65 504. */
int dhaug_split_f16(const float* src, int64_t ld_src, uint16_t* dst, int64_t rows, int64_t cols, int64_t pad_cols, int mode,
                    void* stream);

/* c_f32[M,N] = act( A[M,K] * B[N,K]^T + bias + residual_f32 ) on IEEE-half operands (v_mfma_f32_16x16x32_f16, fp32 accumulate):
 * with A = dhaug_split_f16(x, mode 0) and B = dhaug_split_f16(W, mode 1), K = 3 * padded width, this is the layer
 * x W^T in the "f16x3" arithmetic of the fused parity programs (Whi Xhi + Whi Xlo + Wlo Xhi: logits <= 1e-4 rel of the fp32
 * reference, R/models_Fk_GAN/Fk_discriminator.py:180-201, 253-266, 381-587) at ANY width -- the parity-grade forward at the
 * reference's default DenseDim 1000 (R/function_aug/config.py:101-109), where the fused programs do not apply.  Runs on the
 * 256 x 256 x 64 ping-pong tiles (csrc/dhaug_gemm_p8.hip); takes K >= 128, N % 8 == 0, ldc_f32 % 4 == 0 and 16-byte aligned bias /
 * residual / output, DHAUG_EUNSUPPORTED otherwise (the caller then runs that layer in "bf16x6").  Read extent as dhaug_gemm_bf16. */
int dhaug_gemm_f16x3(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias, const float* residual_f32,
                     int64_t ld_res_f32, float* c_f32, int64_t ldc_f32, int64_t M, int64_t N, int64_t K, int act, float slope,
                     void* stream);
/* The same layer with the split LEFT OUT of the chain of layers: the activation side may come as the two distinct pieces [hi|lo] of kp
 * columns each (a_planes != 0: A = dhaug_split_f16(x, mode 2), lda >= 2 kp, kp = 64 * 2^j -- DenseDim 1000 is padded to 1 024 --; the
 * kernel reads piece (0 0 1)[s] for K-segment s of B = dhaug_split_f16(W, mode 1, pad kp); a_planes == 0: A is the mode 0 operand,
 * lda >= 3 kp, any kp), and the result may be written once more as such pieces (c_planes, optional: planes_kp >= N columns per piece,
 * ld_planes >= 2 planes_kp, columns [N, planes_kp) zero) -- bit for bit dhaug_split_f16(c_f32, mode 2, pad planes_kp), from the
 * epilogue's registers: the next layer's operand without a split launch (4 instead of 6 bytes per value, read and written once).
 * Same product terms in the same order as dhaug_gemm_f16x3 on the mode 0 operand of the same padded width: identical results.
 * DHAUG_EUNSUPPORTED where the ping-pong kernel does not take the shape. */
int dhaug_gemm_f16x3_planes(const uint16_t* A, int64_t lda, int a_planes, const uint16_t* B, int64_t ldb, const float* bias,
                            const float* residual_f32, int64_t ld_res_f32, float* c_f32, int64_t ldc_f32, uint16_t* c_planes, int64_t ld_planes,
                            int64_t planes_kp, int64_t M, int64_t N, int64_t kp, int act, float slope, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused multi-layer forward (one launch per network; activations stay in LDS)
 * ---------------------------------------------------------------------------------------------------- */

/* Weight fragments in MFMA A-operand order for dhaug_mlp_forward:
 *   dst[((slice*ksteps + ks)*64 + lane)*8 + j] = bf16( W[32*slice + (lane&31)][k0 + 16*ks + 8*(lane>>5) + j] )
 * (0 outside N x K), ksteps = 4*ceil(K/64) <= 16 (whole 64-wide chunks), slice = 0..7 (always 8 slices = 256
 * feature rows, zero beyond N): dst holds 8 * ksteps * 512 bf16.  W is the fp32 nn.Linear weight
 * [N, ldw]; (k0, K) selects a column range (layers fed by a concatenation take one block per source). */
int dhaug_pack_wfrag(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, void* stream);

/* One unit of a fused network program.  Three LDS activation buffers exist per 128-row batch tile:
 * ids 0 and 1 hold up to 256 bf16 columns, id 2 up to 128. */
#define DHAUG_MLP_LOAD_F32    0   /* global fp32 (M, ld) columns [0, cols) -> buffer dst (rounded to bf16)          */
#define DHAUG_MLP_LOAD_BF16   1   /* global bf16 (M, ld) columns [0, cols) -> buffer dst                              */
#define DHAUG_MLP_STORE_BF16  2   /* buffer src columns [0, cols) -> global bf16 (M, ld)                              */
#define DHAUG_MLP_GEMM        3   /* one layer: dst = act(W * src [+ W2 * src2] + bias + res)                          */
#define DHAUG_MLP_LOAD_KCS    5   /* dhaug_mlp_forward_x3 only: global fp32 poses (M, ld >= 48) -> their 30 KCS features  */
                                  /* (special_KCS_Input_transform, R/models_Fk_GAN/Fk_discriminator.py:36-146) in buffer   */
                                  /* dst, columns 30..63 zero: the 3D critic's KCS branch without a launch of its own        */
#define DHAUG_MLP_F_OUT_F32   4   /* network output (n <= 64): fp32 (M, ld) to g, staged through buffer dst (0 or 1)    */
#define DHAUG_MLP_F_DOT_OUT  16   /* GEMM whose only consumer is a 1-wide linear layer (a critic's logit): the activation  */
                                  /* is not stored; its dot product with w2 = fp32 [257] (that layer's weights as bf16      */
                                  /* values, zero beyond n, bias at [256]) goes to g (M, ld) column 0 as fp32; dst (0 or 1,  */
                                  /* not src / res) is scratch for the partial sums                                          */
typedef struct dhaug_mlp_unit {
    int kind, flags;
    int src, dst, res;            /* buffer ids, -1 = none; dst may equal res (in-place residual), never src        */
    int src2, ksteps2;            /* optional second source (input = concatenation of two buffers); ksteps2 = 0: none */
    int ksteps;                   /* K/16 of the first source (K zero-padded to 16 by the producer)                  */
    int n;                        /* output features of the layer (<= 256)                                          */
    int act;                      /* DHAUG_ACT_*                                                                     */
    float slope;
    int cols;                     /* LOAD/STORE: columns moved (multiple of 8)                                       */
    int64_t ld;
    const void* g;                /* global tensor of LOAD / STORE / OUT_F32                                         */
    const void* w;                /* GEMM: packed fragments of the first source's weight columns (dhaug_pack_wfrag)  */
    const void* w2;               /* fragments of the second source's weight columns                                 */
    const float* bias;            /* GEMM: fp32 [256], zero padded                                                   */
    void* save;                   /* GEMM (dhaug_mlp_forward, not OUT_F32 / DOT_OUT), optional: the layer's output also   */
    int64_t save_ld;              /* goes to global memory as bf16 (M, save_ld), columns [0, ceil16(n)) (zero beyond n):   */
                                  /* forward-with-save of the training step (the saved activations of                     */
                                  /* R/models_Fk_GAN/model_fk_gan_train.py:177-230's autograd graph)                      */
    void* bits;                   /* GEMM with `save`, a full-width (n > 224) layer INSIDE a run of such layers, optional:   */
                                  /* one bit per output element, (y > 0) -- the mask act'(y) of the backward and tangent       */
                                  /* sweeps in 1/16 of the bytes.  uint32 [ceil(M/128)*4][4][64]: word ((T*4 + w)*64 + l) of  */
                                  /* 32-row tile T covers row 32 T + (l & 31), features 32 (w + 4 t) + 8 g + 4 (l >> 5) + e;   */
                                  /* element j = 16 t + 4 g + e sits at bit j/2 (j even) or 16 + j/2 (j odd).  Consumed by    */
                                  /* dhaug_gemm_bf16_dbits.  A unit that is not such a layer -> DHAUG_EUNSUPPORTED.          */
    int64_t save_rows;            /* with `save`: 0 = every row is saved; n > 0 = rows [0, n) only; < 0 = none (the sign bits are   */
                                  /* written for every row regardless).  The interpolated rows of a critic step's batch are read   */
                                  /* back by nothing but their masks: R/models_Fk_GAN/Fk_discriminator.py:205-231 needs D(x_hat)'s  */
                                  /* input gradient, not its activations.                                                          */
} dhaug_mlp_unit;

/* dhaug_pack_wfrag for every layer of a network in one launch (after each optimizer step of a training loop that runs the
 * fused programs: R/models_Fk_GAN/model_fk_gan_train.py:177-230 steps the critics every iteration).  Per descriptor: the
 * fragment blob of W[:, k0:k0+K] (ksteps = K padded to multiples of 64, in units of 16; always 8 slices), the zero-padded
 * fp32 bias [256] (bias_dst, optional) and, for a 1-wide logit layer folded into its producer, the DOT_OUT vector [257]
 * (dot_dst, optional).  descs_device: array of n descriptors in device memory. */
typedef struct dhaug_wfrag_desc {
    const float* W;
    int64_t ldw;
    void* dst;
    const float* bias;
    float* bias_dst;
    float* dot_dst;
    int32_t N, K, k0, ksteps;
} dhaug_wfrag_desc;
int dhaug_pack_wfrag_batch(const dhaug_wfrag_desc* descs_device, int n, void* stream);

/* Runs the program on every 128-row tile of the batch: replaces the nn.Sequential / myResNet stacks of
 * R/models_Fk_GAN/Fk_generator.py:115-119, Fk_discriminator.py:180-201 and :253-266 (inference / sampling passes).
 * bf16 operands, fp32 accumulate, bf16 activations between layers -- the same arithmetic as a chain of
 * dhaug_gemm_bf16 calls. */
int dhaug_mlp_forward(const dhaug_mlp_unit* units, int nunits, int64_t M, void* stream);

/* Parity-grade variant of the same programs: every operand is an fp16 pair x = hi + lo (22 mantissa bits), a product is
 * Whi Xhi + Whi Xlo + Wlo Xhi on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- fp32-grade results (the reference's
 * layers are fp32 nn.Linear: R/models_Fk_GAN/Fk_discriminator.py:180-201,253-266, R/models_Fk_GAN/Fk_generator.py:115-119;
 * tolerance 1e-4 relative on the logits) at three matrix instructions per k-step.  128-row batch tiles; the activation lives
 * in ONE hi / lo image in LDS that every layer updates in place, so the three buffer ids of a program are virtual: the
 * library checks that every value a unit reads is where the kernel can find it (the image for a source, the workspace for a
 * residual) and returns DHAUG_EUNSUPPORTED otherwise.  Values that a later unit adds as a RESIDUAL wait in a global workspace
 * (fp32, written and read back by the same lane): a program with residuals passes g = a 16-byte aligned buffer of
 * DHAUG_MLP_X3_WORKSPACE_BYTES, shared with no concurrent launch, in its GEMM units that are not outputs (any content).
 * Differences from dhaug_mlp_forward: weights come from dhaug_pack_wfrag_f16x2; only LOAD_F32 (cols and ld even, 8-byte
 * aligned base), LOAD_KCS and GEMM units; no second source; no F_DOT_OUT (a 1-wide logit layer is an OUT_F32 unit).
 * dhaug_pack_wfrag_f16x2: dst[(((slice*ksteps + ks)*2 + piece)*64 + lane)*8 + j] = piece(W[32 slice + (lane&31)][k0 + 16 ks +
 * 8 (lane>>5) + j]), piece 0 = fp16(w), piece 1 = fp16(w - piece 0); 8 slices, k-steps padded to multiples of 4, i.e.
 * 2 * 8 * ksteps * 512 fp16 values. */
int dhaug_pack_wfrag_f16x2(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, void* stream);
/* The same pair of pieces in the fragment order of v_mfma_f32_16x16x32_f16 (same size): per slice of 32 features two feature
 * tiles of 16, k-steps of 32 -- dst[((((slice*2 + ft)*(ksteps/2) + ks)*2 + piece)*64 + lane)*8 + j] = piece(W[32 slice + 16 ft +
 * (lane&15)][k0 + 32 ks + 8 (lane>>4) + j]).  A program whose GEMM units carry DHAUG_MLP_F_T16 (all of them or none) runs its
 * layers on that instruction: same image, same results to rounding (k is summed 32 at a time), and -- measured, MI355X,
 * tools/ubench/mfma_shape.hip -- 1.19 x the FLOP/s of the 32 x 32 x 16 form at equal cycles, because the chip holds a higher
 * clock on it. */
#define DHAUG_MLP_F_T16      32
int dhaug_pack_wfrag_f16x2_t16(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, void* stream);
#define DHAUG_MLP_X3_WORKSPACE_BYTES (2 * 256 * 4 * 64 * 128 * 4)   /* regions x workgroups x waves x lanes x values x 4 */
int dhaug_mlp_forward_x3(const dhaug_mlp_unit* units, int nunits, int64_t M, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Elementwise / reductions used by the training step
 * ---------------------------------------------------------------------------------------------------- */

/* column sums: src (M, N) fp32|bf16 -> dst[N] fp32 (bias gradients). */
int dhaug_colsum_f32(const float* src, int64_t ld, float* dst, int64_t M, int64_t N, int accumulate,
                     void* stream);
int dhaug_colsum_bf16(const uint16_t* src, int64_t ld, float* dst, int64_t M, int64_t N, int accumulate,
                      void* stream);

/* dst = g * act'(y)  (y = the saved activation output; relu: y>0, lrelu: y>0 ? 1 : slope), bf16 in/out,
 * optional second output g_res (same values) is the gradient flowing into a residual branch. */
int dhaug_act_backward_bf16(const uint16_t* g, int64_t ld_g, const uint16_t* y, int64_t ld_y,
                            uint16_t* dst, int64_t ld_dst, int64_t M, int64_t N, int act, float slope,
                            void* stream);

/* fp32 variant on flat arrays of n elements. */
int dhaug_act_backward_f32(const float* g, const float* y, float* dst, int64_t n, int act, float slope,
                           void* stream);

/* Fused Adam step on a flat fp32 parameter vector (torch.optim.Adam semantics, eps outside the sqrt,
 * bias correction; R/models_Fk_GAN/model_fk_gan_train.py:112-118: lr 1e-4, betas (0.5, 0.9)).
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce sum). */
int dhaug_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* The same step with the step count in device memory (*step_dev >= 1, advanced with dhaug_counter_add before the call): a
 * captured hipGraph replays the launch, so the bias corrections must not be baked into kernel arguments. */
int dhaug_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, const int* step_dev, float grad_scale, void* stream);
int dhaug_counter_add(int* counter, int value, void* stream);

/* The bf16 operand copies of ALL weights of a network in two launches (after its Adam step): per weight W (N, K) fp32,
 * contiguous: nt = bf16 (N, Kp) zero-padded along K (operand of x W^T), nn = bf16 (K, Np) = W^T zero-padded along N (operand
 * of g W); Kp, Np multiples of 16.  descs_device: array of nparams descriptors in device memory. */
typedef struct dhaug_repack_desc {
    const float* W;
    void* nt;
    void* nn;
    int N, K, Kp, Np;
} dhaug_repack_desc;
int dhaug_repack_weights(const dhaug_repack_desc* descs_device, int nparams, void* stream);

/* The optimizer step of a whole network as two streaming launches (replaces dhaug_counter_add + dhaug_adam_step_dev +
 * dhaug_repack_weights; same arithmetic, torch.optim.Adam.step at R/models_Fk_GAN/model_fk_gan_train.py:218,470): Adam on the flat
 * vectors with the bf16 copy nt of every weight written from the updated values, then nn = nt^T.
 * descs_device (device memory, ndesc entries, ascending item0, together covering every element of the flat vectors exactly once):
 * a weight W (N, K) at element offset `off` with its copy nt (N, Kp) as in dhaug_repack_desc, items = ceil(N * Kp / 4096); or
 * N = 0: `len` plain elements at `off`, items = ceil(len / 4096).  nitems = the sum.  weights_device: the dhaug_repack_desc of the
 * same weights (nweights of them), for the transposed copies.  state: the step count in device memory; the call uses count + 1
 * for the bias corrections and leaves count + 1 there -- a captured hipGraph replays it. */
typedef struct dhaug_adam_desc {
    long long off, len;
    void* nt;
    int N, K, Kp, pad_;
    long long item0;
} dhaug_adam_desc;
int dhaug_adam_repack_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                           float eps, int* state, float grad_scale, const dhaug_adam_desc* descs_device, int ndesc, int64_t nitems,
                           const dhaug_repack_desc* weights_device, int nweights, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * WGAN-GP critic step arithmetic (R/models_Fk_GAN/Fk_discriminator.py:205-231, model_fk_gan_train.py:186-221)
 * ---------------------------------------------------------------------------------------------------- */

/* out (3B, W) fp32: rows [0,B) = real, [B,2B) = fake, [2B,3B) = alpha_b * real + (1 - alpha_b) * fake -- the batch one
 * critic step scores (the two critic passes and the penalty's interpolates as one batch). */
int dhaug_gp_assemble(const float* real, const float* fake, const float* alpha, float* out, int64_t B, int64_t W, void* stream);
/* ... and rows [0, 2B) (real, fake) once more as bf16 (2B, ld_bf16 >= W; columns beyond W are not written): the operand the step's
 * weight-gradient sweep contracts the input layer's cotangent with, without a cast launch of its own (same rounding: nearest even) */
int dhaug_gp_assemble_bf16(const float* real, const float* fake, const float* alpha, float* out, uint16_t* rows_bf16, int64_t ld_bf16,
                           int64_t B, int64_t W, void* stream);

/* Per row b of grad (B, W) = dD/dx_hat:  n = ||grad_b||_2;  pen[b] = (n - 1)^2;  v_b = coef * (n - 1) / n * grad_b, the
 * cotangent of the penalty on grad (coef = 2 * LAMBDA / B gives d/dgrad of LAMBDA * mean((n - 1)^2)). */
int dhaug_gp_penalty(const float* grad, float* v, float* pen, int64_t B, int64_t W, float coef, void* stream);
/* ... with v also as bf16 (B, ld_bf16 >= W): the tangent sweep's first operand */
int dhaug_gp_penalty_bf16(const float* grad, float* v, uint16_t* v_bf16, int64_t ld_bf16, float* pen, int64_t B, int64_t W, float coef,
                          void* stream);

/* Frame differences of clips (the motion critics' diff branches, R/models_Fk_GAN/Fk_discriminator.py:458-460,489-492,570-573):
 * x (rows, R*in_w) -> out (rows, (R-1)*w), out[r][f][c] = x[r][f+1][c] - x[r][f][c] over the first w columns of every frame;
 * adjoint != 0: the transposed map, x (rows, (R-1)*w) -> out (rows, R*in_w) (zero for columns >= w). */
int dhaug_frame_diff(const float* x, float* out, int64_t rows, int R, int in_w, int w, int adjoint, void* stream);

/* Frame reversal of clips, x (rows, R*w) -> out[r][f][:] = x[r][R-1-f][:] (torch.flip(dims=[1]) of the video loop's playback
 * copies, R/models_Fk_GAN/video_GAN_fun.py:467,521); the map is its own transpose, so the same call back-propagates. */
int dhaug_frame_reverse(const float* x, float* out, int64_t rows, int R, int w, void* stream);

/* out[0] = sum_i weights[i] * mean(arrays[i][0 .. counts[i])): the generator loss gen_loss (or -gen_loss) from the critics'
 * logit arrays in one launch (R/models_Fk_GAN/model_fk_gan_train.py:470-476).  arrays / counts / weights: HOST arrays of
 * n <= DHAUG_WEIGHTED_MEANS_MAX entries (read during the call); the logit arrays and out are device memory. */
#define DHAUG_WEIGHTED_MEANS_MAX 16
int dhaug_weighted_means(const float* const* arrays, const int64_t* counts, const float* weights, int n, float* out, void* stream);

/* out5 = { D_real, D_fake, GP = lambda * mean(pen), Wasserstein_D = D_real - D_fake, D_cost = D_fake - D_real + GP } from
 * the logits (rows [0,B) real, [B,2B) fake, stride ld) and the P per-row penalties (P = B, or B * frames for the 2D motion
 * critic, whose penalty is taken per frame). */
int dhaug_critic_scalars(const float* logits, int64_t ld, const float* pen, int64_t B, int64_t P, float lambda, float* out5,
                         float* scratch /* DHAUG_CRITIC_SCALARS_SCRATCH floats, any content */, void* stream);
#define DHAUG_CRITIC_SCALARS_SCRATCH 192

/* The TOP of a branch critic -- merge layer (concatenation of two 256-wide branches -> n0 <= 112 features), one n0-wide myResNet
 * block, the 1-wide logit layer (R/models_Fk_GAN/Fk_discriminator.py:149-201: merge_previous, merge_block1, output) -- in the
 * explicit training step, ONE launch per sweep; the n0-wide tensors stay in LDS between the layers, results bit-identical to the
 * launches replaced (dhaug_rank1_mask_bf16 + three dhaug_gemm_bf16_dmask_pad / dhaug_gemm_bf16_dbits_wide calls).
 *   dhaug_critic_top_backward_bf16 (sweep 2):
 *     g2 = bf16(seed w_out) * act'(m1);  g1 = (g2 W2^T) * act'(mh);  g0 = (g1 W1^T + g2) * act'(m0);  gcat = (g0 Wm^T) * act'(cat)
 *     with W2 / W1 (n0, >= 112) and Wm (512, >= 112) the operand copies whose ROWS are the product's outputs (the "nn" copies of
 *     merge_block1.fc2 / fc1 and merge_previous), the masks of the concatenation as the two sign-bit arrays of its 256-column blocks
 *     (struct dhaug_mlp_unit.bits).  m1, mh, m0: the saved activations (M, ld_m) bf16; g2, g1, g0 (M, ld_g) bf16 with zero columns
 *     [n0, 112); gcat (M, ld_gcat >= 512).  M a multiple of 64 (DHAUG_EUNSUPPORTED otherwise: the caller uses the separate launches).
 * All matrices bf16 with 16-byte aligned rows. */
typedef struct dhaug_top_desc {
    const uint16_t* seed; int64_t ld_seed;              /* (M, >= 1): column 0 = the logit cotangent of the row */
    const uint16_t* wout; int64_t ld_wout;              /* the logit layer's n0 weights, element stride ld_wout */
    const uint16_t* x; int64_t ldx;                     /* tangent sweep: the tangent of the concatenation (M, 512) */
    const uint16_t* m1; const uint16_t* mh; const uint16_t* m0; int64_t ld_m;
    const uint16_t* w2; int64_t ldw2;
    const uint16_t* w1; int64_t ldw1;
    const uint16_t* wm; int64_t ldwm;
    const uint32_t* bits0; const uint32_t* bits1;
    uint16_t* g2; uint16_t* g1; uint16_t* g0; int64_t ld_g;
    uint16_t* gcat; int64_t ld_gcat;
    int64_t M; int64_t n0; int64_t nc;                  /* nc = 512 */
    int32_t mask_act; float mask_slope;
} dhaug_top_desc;
int dhaug_critic_top_backward_bf16(const dhaug_top_desc* d, void* stream);
/*   dhaug_critic_top_tangent_bf16 (sweep 3, rows = the interpolated rows): IN PLACE over the rows of the saved activations,
 *     m0 <- (x Wm^T) * act'(m0);  mh <- (m0' W1^T) * act'(mh);  m1 <- (mh' W2^T + m0') * act'(m1)      (primes: the values just written)
 *     x (M, ldx >= 512) the tangent of the concatenation; Wm (n0, >= 512), W1 / W2 (n0, >= 112) the "nt" operand copies (rows = the
 *     layer's outputs).  Replaces one dhaug_gemm_bf16_dmask_pad with K = 512 and two with K = 112 (the second with the skip); same bits. */
int dhaug_critic_top_tangent_bf16(const dhaug_top_desc* d, void* stream);

/* The 3D critic's gradient penalty between its backward chain and its tangent sweep in ONE launch (R/models_Fk_GAN/Fk_discriminator.py:
 * 205-231 with the KCS transform of :81-140): g = KCS^T(pose) grad_kcs + grad_pose = dD/dx_hat (N, 48); pen[b] = (||g_b|| - 1)^2;
 * v_b = coef (||g_b|| - 1) / ||g_b|| g_b (0 where the norm is 0); tan_kcs = dKCS(pose)[v].  grad_kcs (N, 30): cosines then lengths;
 * outputs: tan_kcs_bf16 (N, 32) and tan_pose_bf16 = v (N, 48) as the bf16 operands of the tangent sweep's first layers, pen (N) fp32.
 * The operations of dhaug_kcs_backward + dhaug_add_f32 + dhaug_gp_penalty + dhaug_kcs_jvp + two dhaug_cast_pad_bf16 calls in their order (fp32
 * results equal to the last bit or two: the compiler contracts multiply-adds per kernel). */
int dhaug_d3_penalty(const float* pose16, const float* grad_kcs, const float* grad_pose, float coef, uint16_t* tan_kcs_bf16,
                     uint16_t* tan_pose_bf16, float* pen, int64_t N, void* stream);

/* First step of a critic's backward chain, through its 1-wide logit layer: out[r][c] = bf16(seed[r] * w[c]) * act'(mask[r][c])
 * for c < N, zero in [N, pad_cols) -- (gz W_out) * act'(y) of R/models_Fk_GAN/Fk_discriminator.py:201,266's backward, which as
 * a GEMM has K = 1.  seed: bf16, one value per row (stride ld_seed); w: the layer's N weights as bf16 (stride ld_w); mask, out:
 * bf16 (M, ld) with 16-byte aligned rows; pad_cols <= 1024 (DHAUG_EUNSUPPORTED beyond). */
int dhaug_rank1_mask_bf16(const uint16_t* seed, int64_t ld_seed, const uint16_t* w, int64_t ld_w, const uint16_t* mask, int64_t ld_mask,
                          uint16_t* out, int64_t ld_out, int64_t M, int64_t N, int64_t pad_cols, int mask_act, float mask_slope,
                          void* stream);
/* The same for a 256-wide hidden layer whose mask is a sign-bit array (struct dhaug_mlp_unit.bits): no mask image is read. */
int dhaug_rank1_bits_bf16(const uint16_t* seed, int64_t ld_seed, const uint16_t* w, int64_t ld_w, const uint32_t* bits,
                          uint16_t* out, int64_t ld_out, int64_t M, int mask_act, float mask_slope, void* stream);

/* out = a + b over n fp32 values (the branch contributions to dD/dx_hat of a multi-branch critic). */
int dhaug_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DHAUG_H */
