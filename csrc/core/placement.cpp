#include "placement.hpp"

#include <algorithm>
#include <map>
#include <numeric>

namespace lc {

namespace {

struct Unit {  // one set of weights (members with equal model_key collapse into one unit)
  std::string key;
  std::vector<int> member_idx;
  int group = 1;
  int64_t bytes_per_gpu = 0;
};

// First-fit-decreasing of units onto `g` GPUs with capacity `cap`. Multi-GPU units
// take `group` consecutive empty-most GPUs. Returns false if it does not fit.
bool pack_units(const std::vector<Unit>& units, int g, int64_t cap, bool one_unit_per_gpu,
                std::vector<std::vector<int>>* unit_devices, std::vector<int64_t>* used) {
  used->assign(g, 0);
  unit_devices->assign(units.size(), {});
  std::vector<int> order(units.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (units[a].group != units[b].group) return units[a].group > units[b].group;
    return units[a].bytes_per_gpu > units[b].bytes_per_gpu;
  });
  std::vector<int> occupancy(g, 0);
  for (int ui : order) {
    const Unit& u = units[ui];
    if (u.group > g) return false;
    // candidate GPUs sorted by (occupancy, used bytes)
    std::vector<int> gpus(g);
    std::iota(gpus.begin(), gpus.end(), 0);
    std::stable_sort(gpus.begin(), gpus.end(), [&](int a, int b) {
      if (occupancy[a] != occupancy[b]) return occupancy[a] < occupancy[b];
      return (*used)[a] < (*used)[b];
    });
    std::vector<int> chosen;
    for (int d : gpus) {
      if ((*used)[d] + u.bytes_per_gpu > cap) continue;
      if (one_unit_per_gpu && occupancy[d] > 0) continue;
      chosen.push_back(d);
      if (static_cast<int>(chosen.size()) == u.group) break;
    }
    if (static_cast<int>(chosen.size()) < u.group) return false;
    std::sort(chosen.begin(), chosen.end());
    for (int d : chosen) {
      (*used)[d] += u.bytes_per_gpu;
      ++occupancy[d];
    }
    (*unit_devices)[ui] = chosen;
  }
  return true;
}

}  // namespace

PlacementPlan plan_placement(const std::vector<MemberSpec>& members, int num_gpus,
                             int64_t hbm_bytes_per_gpu, double reserve_fraction,
                             PlacementPolicy policy) {
  PlacementPlan plan;
  if (members.empty() || num_gpus <= 0) {
    plan.error = "no members or no GPUs";
    return plan;
  }
  const int64_t cap = static_cast<int64_t>(hbm_bytes_per_gpu * (1.0 - reserve_fraction));

  std::vector<Unit> units;
  std::map<std::string, int> unit_of;
  for (size_t i = 0; i < members.size(); ++i) {
    const MemberSpec& m = members[i];
    const int group = std::max(1, std::max(m.tp, m.ep));
    std::string key = m.model_key.empty() ? ("\x01" + m.name) : m.model_key;
    auto it = unit_of.find(key);
    if (it == unit_of.end()) {
      Unit u;
      u.key = key;
      u.group = group;
      u.bytes_per_gpu = (m.weight_bytes + group - 1) / group + (m.min_kv_bytes + group - 1) / group;
      u.member_idx.push_back(static_cast<int>(i));
      unit_of[key] = static_cast<int>(units.size());
      units.push_back(std::move(u));
    } else {
      Unit& u = units[it->second];
      if (group != u.group) {
        plan.error = "members sharing model_key '" + m.model_key + "' disagree on tp/ep";
        return plan;
      }
      // personas on a shared model add KV demand, not weights
      u.bytes_per_gpu += (m.min_kv_bytes + group - 1) / group;
      u.member_idx.push_back(static_cast<int>(i));
    }
  }

  int g_min = -1;
  std::vector<std::vector<int>> unit_devices;
  std::vector<int64_t> used;
  int spread_target = 0;
  for (const auto& u : units) spread_target += u.group;

  auto try_g = [&](int g, bool exclusive) { return pack_units(units, g, cap, exclusive, &unit_devices, &used); };

  if (policy == PlacementPolicy::Spread && spread_target <= num_gpus && try_g(spread_target, true)) {
    g_min = spread_target;
  } else {
    for (int g = 1; g <= num_gpus; ++g)
      if (try_g(g, false)) {
        g_min = g;
        break;
      }
  }
  if (g_min < 0) {
    plan.error = "ensemble does not fit in " + std::to_string(num_gpus) + " GPUs";
    // Name what can never fit, whatever its neighbours: a model whose group is larger than the
    // node, or whose per-GPU share alone exceeds what a GPU offers.
    for (const auto& u : units) {
      const std::string& who = members[u.member_idx[0]].name;
      if (u.group > num_gpus) {
        plan.error += ": '" + who + "' wants a group of " + std::to_string(u.group) + " GPUs";
        break;
      }
      if (u.bytes_per_gpu > cap) {
        plan.error += ": '" + who + "' needs " + std::to_string(u.bytes_per_gpu >> 20) + " MiB per GPU (weights + KV), " +
                      std::to_string(cap >> 20) + " MiB are available; shard it further (tp / ep)";
        break;
      }
    }
    return plan;
  }

  plan.feasible = true;
  plan.gpus_per_replica = g_min;
  const int n_rep = num_gpus / g_min;
  plan.bytes_per_device.assign(num_gpus, 0);
  for (int r = 0; r < n_rep; ++r) {
    ReplicaPlan rp;
    const int base = r * g_min;
    for (int d = 0; d < g_min; ++d) {
      rp.devices.push_back(base + d);
      plan.bytes_per_device[base + d] = used[d];
    }
    rp.members.resize(members.size());
    for (size_t ui = 0; ui < units.size(); ++ui) {
      bool first = true;
      for (int mi : units[ui].member_idx) {
        MemberPlacement mp;
        mp.name = members[mi].name;
        for (int d : unit_devices[ui]) mp.devices.push_back(base + d);
        mp.shares_weights = !first;
        first = false;
        rp.members[mi] = std::move(mp);
      }
    }
    plan.replicas.push_back(std::move(rp));
  }
  for (int d = n_rep * g_min; d < num_gpus; ++d) plan.idle_devices.push_back(d);
  return plan;
}

}  // namespace lc
