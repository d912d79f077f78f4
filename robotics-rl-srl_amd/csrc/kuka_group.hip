// kuka_group.hip — lane-group Kuka rollout kernels for the baked model (see kuka_group.hpp / kuka_group_kernels.hpp).
#include "kuka_group_kernels.hpp"

namespace srl {
using namespace kuka;

SRL_GROUP_LAUNCHER(kuka_group_launch_baked, false)

}  // namespace srl
