// Member -> GPU placement for one node.
//
// The reference has no placement problem: every persona is the same remote model
// (/root/reference/src/main.rs:82-86). Here each ensemble member is a model resident
// in HBM, possibly tensor- or expert-parallel over several GPUs, and whole ensembles
// are replicated over spare GPUs for request-level data parallelism.
//
// The planner works in units of "ensemble replicas": a replica is a set of GPUs that
// together host every member once. With one process per GPU, rank r serves the
// members whose device list contains r, and questions are sharded over replicas.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lc {

struct MemberSpec {
  std::string name;
  std::string model_key;       // members with equal keys share one set of weights (personas)
  int64_t weight_bytes = 0;    // unsharded parameter bytes
  int64_t min_kv_bytes = 0;    // KV pool the member needs at least
  int tp = 1;                  // tensor-parallel degree
  int ep = 1;                  // expert-parallel degree (MoE); group size = max(tp, ep)
};

enum class PlacementPolicy : uint8_t {
  Pack = 0,    // fewest GPUs per replica (co-locate members while HBM allows) -> most replicas
  Spread = 1,  // one member group per GPU set when GPUs allow (config "one model per GPU")
};

struct MemberPlacement {
  std::string name;
  std::vector<int> devices;    // GPUs (node-local ids) of this member inside its replica
  bool shares_weights = false; // true if an earlier member with the same model_key owns the weights
};

struct ReplicaPlan {
  std::vector<int> devices;                // GPUs of this replica
  std::vector<MemberPlacement> members;
};

struct PlacementPlan {
  bool feasible = false;
  std::string error;
  int gpus_per_replica = 0;
  std::vector<ReplicaPlan> replicas;
  std::vector<int64_t> bytes_per_device;   // planned weights + min KV, per GPU
  std::vector<int> idle_devices;
};

PlacementPlan plan_placement(const std::vector<MemberSpec>& members, int num_gpus,
                             int64_t hbm_bytes_per_gpu, double reserve_fraction,
                             PlacementPolicy policy);

}  // namespace lc
