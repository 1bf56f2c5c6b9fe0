// Per-caller helper resources for entry points that fork work onto side streams.
//
// The C ABI takes the caller's stream and must behave as if everything ran on it.  An entry point that
// overlaps independent parts on side streams needs helper streams and fork/join events; they must belong
// to ONE caller (device, stream): shared events let caller A's re-record slip between caller B's record and
// wait, and streams created on one device cannot serve another.  This registry hands every key its own
// resource object, created on first use under a mutex, and never hands the same object to two keys.
// It is HIP-free (the resource type brings its own constructor), so the host logic is unit-tested with a
// plain C++ compiler (tests/native/test_stream_registry.cpp).
#pragma once
#include <cstddef>
#include <map>
#include <memory>
#include <mutex>
#include <utility>

namespace mi {

template <typename Key, typename Resource>
class KeyedRegistry {
 public:
  explicit KeyedRegistry(size_t capacity) : capacity_(capacity) {}

  // The resource of `key`, created by make() on first use.  nullptr when the registry is full (the caller then
  // runs unforked) or when make() returned nullptr (resource creation failed; not cached, so a later call retries).
  template <typename Make>
  Resource *get(const Key &key, Make make) {
    std::lock_guard<std::mutex> lock(mu_);
    auto it = items_.find(key);
    if (it != items_.end()) return it->second.get();
    if (items_.size() >= capacity_) return nullptr;
    std::unique_ptr<Resource> r = make();
    if (!r) return nullptr;
    Resource *raw = r.get();
    items_.emplace(key, std::move(r));
    return raw;
  }

  // The resource of `key` if it exists (never creates).
  Resource *find(const Key &key) {
    std::lock_guard<std::mutex> lock(mu_);
    auto it = items_.find(key);
    return it == items_.end() ? nullptr : it->second.get();
  }

  // Drop the resource of `key` (its destructor releases streams / events).  Returns whether one existed.
  bool release(const Key &key) {
    std::lock_guard<std::mutex> lock(mu_);
    return items_.erase(key) > 0;
  }

  size_t size() {
    std::lock_guard<std::mutex> lock(mu_);
    return items_.size();
  }

 private:
  std::mutex mu_;
  std::map<Key, std::unique_ptr<Resource>> items_;
  size_t capacity_;
};

}  // namespace mi
