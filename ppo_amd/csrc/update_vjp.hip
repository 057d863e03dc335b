// ppo_evaluate_backward: the fused forward/backward kernel of update.hip compiled once more in its SEEDED
// form - the gradients w.r.t. evaluate()'s outputs come from the caller instead of from the PPO loss
// (update_common.h: head_loss_grad) - under the kernel name fwdbwd_vjp_kernel.  A translation unit of
// its own, so that the product kernel's code is untouched and the two compile side by side.
#define PPO_FWDBWD_SEEDED 1
#include "update.hip"
