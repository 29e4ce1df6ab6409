// Reader for PyTorch checkpoints (`torch.save` zip archives), the format the reference loads through
// burn-store's PytorchStore (mod.rs:229-233).  Only what a state_dict needs: the archive's stored
// (uncompressed) members are memory-mapped and the pickle is walked with a small interpreter that knows
// OrderedDict, _rebuild_tensor_v2, _rebuild_parameter and the storage persistent ids.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace matrix_eyes {

struct CheckpointError : std::runtime_error {  // LoaderError::Pytorch
    using std::runtime_error::runtime_error;
};

struct PtTensor {
    std::string name;            // state_dict key
    std::string dtype;           // "f16" | "bf16" | "f32" | "f64" | "i64" | ...
    std::vector<int64_t> dims;
    const void* data = nullptr;  // contiguous, inside the mapped file
    size_t nbytes = 0;
};

class PtFile {
public:
    explicit PtFile(const std::string& path);
    ~PtFile();
    PtFile(const PtFile&) = delete;
    PtFile& operator=(const PtFile&) = delete;
    const std::vector<PtTensor>& tensors() const { return tensors_; }

private:
    void* map_ = nullptr;
    size_t size_ = 0;
    std::vector<PtTensor> tensors_;
};

}  // namespace matrix_eyes
