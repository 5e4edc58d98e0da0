// Counters and latency summaries for the serving loop. The reference measures
// nothing (its only timer paces the poll loop, /root/reference/src/main.rs:449-454);
// this engine reports answers/s, latency percentiles, rounds per question, agreement
// rate and token throughput.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace lc {

class LatencySeries {
 public:
  void add(double ms) { v_.push_back(ms); }
  size_t count() const { return v_.size(); }
  double percentile(double p) const {
    if (v_.empty()) return 0.0;
    std::vector<double> s = v_;
    std::sort(s.begin(), s.end());
    double pos = p / 100.0 * (s.size() - 1);
    size_t lo = static_cast<size_t>(pos);
    size_t hi = std::min(lo + 1, s.size() - 1);
    return s[lo] + (s[hi] - s[lo]) * (pos - lo);
  }
  double mean() const {
    if (v_.empty()) return 0.0;
    double t = 0;
    for (double x : v_) t += x;
    return t / v_.size();
  }
  void clear() { v_.clear(); }

 private:
  std::vector<double> v_;
};

class Metrics {
 public:
  void inc(const std::string& name, double by = 1.0) {
    std::lock_guard<std::mutex> g(mu_);
    counters_[name] += by;
  }
  void observe(const std::string& name, double value) {
    std::lock_guard<std::mutex> g(mu_);
    series_[name].add(value);
  }
  double counter(const std::string& name) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = counters_.find(name);
    return it == counters_.end() ? 0.0 : it->second;
  }
  double percentile(const std::string& name, double p) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = series_.find(name);
    return it == series_.end() ? 0.0 : it->second.percentile(p);
  }
  double mean(const std::string& name) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = series_.find(name);
    return it == series_.end() ? 0.0 : it->second.mean();
  }
  size_t count(const std::string& name) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = series_.find(name);
    return it == series_.end() ? 0 : it->second.count();
  }
  std::map<std::string, double> snapshot() const {
    std::lock_guard<std::mutex> g(mu_);
    std::map<std::string, double> out = counters_;
    for (const auto& kv : series_) {
      out[kv.first + ".count"] = static_cast<double>(kv.second.count());
      out[kv.first + ".mean"] = kv.second.mean();
      out[kv.first + ".p50"] = kv.second.percentile(50);
      out[kv.first + ".p95"] = kv.second.percentile(95);
    }
    return out;
  }
  void reset() {
    std::lock_guard<std::mutex> g(mu_);
    counters_.clear();
    series_.clear();
  }

 private:
  mutable std::mutex mu_;
  std::map<std::string, double> counters_;
  std::map<std::string, LatencySeries> series_;
};

}  // namespace lc
