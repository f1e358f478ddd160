// kaldi-decoder_amd/python/csrc/kaldi-decoder.cc -- the _kaldi_decoder module.
//
// Same Python names, methods, keyword arguments and defaults as the reference's
// bindings (/root/reference/kaldi-decoder/python/csrc/*.cc): DecodableInterface
// (subclassable, snake_case overrides), DecodableCtc(feats, offset=0),
// FasterDecoderOptions(beam, max_active, min_active, beam_delta, hash_ratio),
// FasterDecoder(fst, config) with set_options / decode / reached_final /
// get_best_path(use_final_probs=True) -> (bool, lattice) / init_decoding /
// advance_decoding(decodable, max_num_frames=-1) / num_frames_decoded.
// Where the reference takes kaldifst FST objects this module has its own Fst
// and Lattice classes (kaldifst is not part of this build).

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <limits>

#include "decodable-ctc.h"
#include "decodable-itf.h"
#include "faster-decoder.h"
#include "fst-io.h"
#include "fst.h"
#include "lattice-decoder.h"

namespace py = pybind11;
using namespace kaldi_decoder;

namespace {

class PyDecodableInterface : public DecodableInterface {
 public:
  using DecodableInterface::DecodableInterface;
  float LogLikelihood(int32_t frame, int32_t index) override {
    PYBIND11_OVERRIDE_PURE_NAME(float, DecodableInterface, "log_likelihood", LogLikelihood, frame,
                                index);
  }
  bool IsLastFrame(int32_t frame) const override {
    PYBIND11_OVERRIDE_PURE_NAME(bool, DecodableInterface, "is_last_frame", IsLastFrame, frame);
  }
  int32_t NumFramesReady() const override {
    PYBIND11_OVERRIDE_NAME(int32_t, DecodableInterface, "num_frames_ready", NumFramesReady);
  }
  int32_t NumIndices() const override {
    PYBIND11_OVERRIDE_PURE_NAME(int32_t, DecodableInterface, "num_indices", NumIndices);
  }
};

using FArray = py::array_t<float, py::array::c_style | py::array::forcecast>;
using IArray = py::array_t<int32_t, py::array::c_style | py::array::forcecast>;

template <class T>
std::vector<T> ToVec(const py::array_t<T, py::array::c_style | py::array::forcecast> &a) {
  return std::vector<T>(a.data(), a.data() + a.size());
}

py::tuple LatticeArrays(const Lattice &l) {
  const int64_t na = l.NumArcs();
  py::array_t<int32_t> src(na), dst(na), il(na), ol(na);
  py::array_t<float> gc(na), ac(na);
  int64_t k = 0;
  for (int32_t s = 0; s < l.NumStates(); ++s)
    for (const LatticeArc &a : l.Arcs(s)) {
      src.mutable_data()[k] = s;
      dst.mutable_data()[k] = a.nextstate;
      il.mutable_data()[k] = a.ilabel;
      ol.mutable_data()[k] = a.olabel;
      gc.mutable_data()[k] = a.weight.graph;
      ac.mutable_data()[k] = a.weight.acoustic;
      ++k;
    }
  return py::make_tuple(src, dst, il, ol, gc, ac);
}

py::tuple LinearSymbolSequence(const Lattice &l) {
  std::vector<int32_t> is, os;
  LatticeWeight w;
  const bool ok = l.GetLinearSymbolSequence(&is, &os, &w);
  return py::make_tuple(ok, is, os, py::make_tuple(w.graph, w.acoustic));
}

}  // namespace

PYBIND11_MODULE(_kaldi_decoder, m) {
  m.doc() = "MI355X-native kaldi-decoder: pybind11 binding over the libkd_hip C ABI";

  py::class_<DecodableInterface, PyDecodableInterface>(m, "DecodableInterface")
      .def(py::init<>())
      .def("log_likelihood", &DecodableInterface::LogLikelihood, py::arg("frame"), py::arg("index"))
      .def("is_last_frame", &DecodableInterface::IsLastFrame, py::arg("frame"))
      .def("num_frames_ready", &DecodableInterface::NumFramesReady)
      .def("num_indices", &DecodableInterface::NumIndices);

  py::class_<DecodableCtc, DecodableInterface>(m, "DecodableCtc", py::dynamic_attr())
      // copies, like the reference's FloatMatrix constructor
      .def(py::init([](const FArray &feats, int32_t offset) {
             if (feats.ndim() != 2) throw std::runtime_error("feats must be a 2-D float32 array");
             return DecodableCtc::Copy(feats.data(), static_cast<int32_t>(feats.shape(0)),
                                       static_cast<int32_t>(feats.shape(1)), offset);
           }),
           py::arg("feats"), py::arg("offset") = 0)
      // zero-copy view of a row-major float32 matrix already in GPU (or host) memory
      .def_static(
          "from_pointer",
          [](uintptr_t ptr, int32_t num_rows, int32_t num_cols, int32_t offset, bool on_device,
             int64_t row_stride) {
            return new DecodableCtc(reinterpret_cast<const float *>(ptr), num_rows, num_cols,
                                    offset, on_device, row_stride);
          },
          py::arg("ptr"), py::arg("num_rows"), py::arg("num_cols"), py::arg("offset") = 0,
          py::arg("on_device") = true, py::arg("row_stride") = -1,
          py::return_value_policy::take_ownership);

  py::class_<Fst>(m, "Fst")
      .def(py::init([](int32_t start, const IArray &row_ptr, const IArray &ilabel,
                       const IArray &olabel, const FArray &weight, const IArray &nextstate,
                       const FArray &final_cost) {
             return new Fst(start, ToVec<int32_t>(row_ptr), ToVec<int32_t>(ilabel),
                            ToVec<int32_t>(olabel), ToVec<float>(weight),
                            ToVec<int32_t>(nextstate), ToVec<float>(final_cost));
           }),
           py::arg("start"), py::arg("row_ptr"), py::arg("ilabel"), py::arg("olabel"),
           py::arg("weight"), py::arg("nextstate"), py::arg("final"))
      // OpenFst binary "vector" / "const" files over standard arcs (what
      // kaldifst.StdVectorFst.read / StdConstFst.read load)
      .def_static("read", [](const std::string &path) { return ReadFst(path).release(); },
                  py::arg("filename"), py::return_value_policy::take_ownership)
      .def("write",
           [](const Fst &f, const std::string &path, const std::string &t) { WriteFst(f, path, t); },
           py::arg("filename"), py::arg("fst_type") = "vector")
      .def("arrays",
           [](const Fst &f) {
             auto I = [](const std::vector<int32_t> &v) { return py::array_t<int32_t>(v.size(), v.data()); };
             auto F = [](const std::vector<float> &v) { return py::array_t<float>(v.size(), v.data()); };
             return py::make_tuple(I(f.RowPtr()), I(f.Ilabel()), I(f.Olabel()), F(f.Weight()),
                                   I(f.NextState()), F(f.FinalCosts()));
           },
           "(row_ptr, ilabel, olabel, weight, nextstate, final) CSR arrays")
      .def_property_readonly("start", &Fst::Start)
      .def_property_readonly("num_states", &Fst::NumStates)
      .def_property_readonly("num_arcs", &Fst::NumArcs)
      .def("final", &Fst::Final, py::arg("state"));

  py::class_<Lattice>(m, "Lattice")
      .def(py::init<>())
      .def_property_readonly("start", &Lattice::Start)
      .def_property_readonly("num_states", &Lattice::NumStates)
      .def_property_readonly("num_arcs", &Lattice::NumArcs)
      .def("final",
           [](const Lattice &l, int32_t s) {
             const LatticeWeight w = l.Final(s);
             return py::make_tuple(w.graph, w.acoustic);
           },
           py::arg("state"))
      .def("arcs",
           [](const Lattice &l, int32_t s) {
             py::list out;
             for (const LatticeArc &a : l.Arcs(s))
               out.append(py::make_tuple(a.ilabel, a.olabel,
                                         py::make_tuple(a.weight.graph, a.weight.acoustic),
                                         a.nextstate));
             return out;
           },
           py::arg("state"))
      .def("arrays", &LatticeArrays,
           "(src, dst, ilabel, olabel, graph_cost, acoustic_cost) arrays over all arcs")
      .def("get_linear_symbol_sequence", &LinearSymbolSequence,
           "(ok, isymbols, osymbols, (graph_cost, acoustic_cost)) of a linear lattice");
  m.def("get_linear_symbol_sequence", &LinearSymbolSequence, py::arg("lattice"));

  py::class_<FasterDecoderOptions>(m, "FasterDecoderOptions")
      .def(py::init<float, int32_t, int32_t, float, float>(), py::arg("beam") = 16.0,
           py::arg("max_active") = std::numeric_limits<int32_t>::max(),
           py::arg("min_active") = 20, py::arg("beam_delta") = 0.5, py::arg("hash_ratio") = 2.0)
      .def_readwrite("beam", &FasterDecoderOptions::beam)
      .def_readwrite("max_active", &FasterDecoderOptions::max_active)
      .def_readwrite("min_active", &FasterDecoderOptions::min_active)
      .def_readwrite("beam_delta", &FasterDecoderOptions::beam_delta)
      .def_readwrite("hash_ratio", &FasterDecoderOptions::hash_ratio)
      .def("__str__", &FasterDecoderOptions::ToString);

  py::class_<DeviceOptions>(m, "DeviceOptions")
      .def(py::init([](int device, int64_t arena_tokens, int32_t table_slots, int32_t max_frames) {
             DeviceOptions d;
             d.device = device;
             d.arena_tokens = arena_tokens;
             d.table_slots = table_slots;
             d.max_frames = max_frames;
             return d;
           }),
           py::arg("device") = 0, py::arg("arena_tokens") = 0, py::arg("table_slots") = 0,
           py::arg("max_frames") = 0)
      .def_readwrite("device", &DeviceOptions::device)
      .def_readwrite("arena_tokens", &DeviceOptions::arena_tokens)
      .def_readwrite("table_slots", &DeviceOptions::table_slots)
      .def_readwrite("max_frames", &DeviceOptions::max_frames);

  py::class_<FasterDecoder>(m, "FasterDecoder")
      // the graph is borrowed, as in the reference; keep_alive ties its lifetime
      .def(py::init<const Fst &, const FasterDecoderOptions &, const DeviceOptions &>(),
           py::arg("fst"), py::arg("config"), py::arg("device_options") = DeviceOptions(),
           py::keep_alive<1, 2>())
      .def("set_options", &FasterDecoder::SetOptions, py::arg("config"))
      .def("decode", &FasterDecoder::Decode, py::arg("decodable"))
      .def("reached_final", &FasterDecoder::ReachedFinal)
      .def(
          "get_best_path",
          [](FasterDecoder &self, bool use_final_probs) {
            Lattice lat;
            const bool ok = self.GetBestPath(&lat, use_final_probs);
            return std::make_pair(ok, std::move(lat));
          },
          py::arg("use_final_probs") = true)
      .def("init_decoding", &FasterDecoder::InitDecoding)
      .def("advance_decoding", &FasterDecoder::AdvanceDecoding, py::arg("decodable"),
           py::arg("max_num_frames") = -1)
      .def("num_frames_decoded", &FasterDecoder::NumFramesDecoded);

  // ref: python/csrc/simple-decoder.cc:15-37
  py::class_<SimpleDecoder>(m, "SimpleDecoder")
      .def(py::init<const Fst &, float, const DeviceOptions &>(), py::arg("fst"), py::arg("beam"),
           py::arg("device_options") = DeviceOptions(), py::keep_alive<1, 2>())
      .def("decode", &SimpleDecoder::Decode, py::arg("decodable"))
      .def("reached_final", &SimpleDecoder::ReachedFinal)
      .def(
          "get_best_path",
          [](SimpleDecoder &self, bool use_final_probs) {
            Lattice lat;
            const bool ok = self.GetBestPath(&lat, use_final_probs);
            return std::make_pair(ok, std::move(lat));
          },
          py::arg("use_final_probs") = true)
      .def("final_relative_cost", &SimpleDecoder::FinalRelativeCost)
      .def("init_decoding", &SimpleDecoder::InitDecoding)
      .def("advance_decoding", &SimpleDecoder::AdvanceDecoding, py::arg("decodable"),
           py::arg("max_num_frames") = -1)
      .def("num_frames_decoded", &SimpleDecoder::NumFramesDecoded);

  // ---- lattice decoders (ref: python/csrc/lattice-simple-decoder.cc:15-68) ----
  py::class_<LatticeSimpleDecoderConfig>(m, "LatticeSimpleDecoderConfig")
      .def(py::init<float, float, int32_t, bool, bool, float, float>(), py::arg("beam") = 16.0,
           py::arg("lattice_beam") = 10.0, py::arg("prune_interval") = 25,
           py::arg("determinize_lattice") = true, py::arg("prune_lattice") = true,
           py::arg("beam_ratio") = 0.9, py::arg("prune_scale") = 0.1)
      .def_readwrite("beam", &LatticeSimpleDecoderConfig::beam)
      .def_readwrite("lattice_beam", &LatticeSimpleDecoderConfig::lattice_beam)
      .def_readwrite("prune_interval", &LatticeSimpleDecoderConfig::prune_interval)
      .def_readwrite("determinize_lattice", &LatticeSimpleDecoderConfig::determinize_lattice)
      .def_readwrite("prune_lattice", &LatticeSimpleDecoderConfig::prune_lattice)
      .def_readwrite("beam_ratio", &LatticeSimpleDecoderConfig::beam_ratio)
      .def_readwrite("prune_scale", &LatticeSimpleDecoderConfig::prune_scale)
      .def("__str__", &LatticeSimpleDecoderConfig::ToString);

  py::class_<LatticeFasterDecoderConfig>(m, "LatticeFasterDecoderConfig")
      .def(py::init<float, int32_t, int32_t, float, int32_t, bool, float, float, float, int32_t,
                    int32_t>(),
           py::arg("beam") = 16.0, py::arg("max_active") = std::numeric_limits<int32_t>::max(),
           py::arg("min_active") = 200, py::arg("lattice_beam") = 10.0,
           py::arg("prune_interval") = 25, py::arg("determinize_lattice") = true,
           py::arg("beam_delta") = 0.5, py::arg("hash_ratio") = 2.0, py::arg("prune_scale") = 0.1,
           py::arg("memory_pool_tokens_block_size") = 1 << 8,
           py::arg("memory_pool_links_block_size") = 1 << 8)
      .def_readwrite("beam", &LatticeFasterDecoderConfig::beam)
      .def_readwrite("max_active", &LatticeFasterDecoderConfig::max_active)
      .def_readwrite("min_active", &LatticeFasterDecoderConfig::min_active)
      .def_readwrite("lattice_beam", &LatticeFasterDecoderConfig::lattice_beam)
      .def_readwrite("prune_interval", &LatticeFasterDecoderConfig::prune_interval)
      .def_readwrite("determinize_lattice", &LatticeFasterDecoderConfig::determinize_lattice)
      .def_readwrite("beam_delta", &LatticeFasterDecoderConfig::beam_delta)
      .def_readwrite("hash_ratio", &LatticeFasterDecoderConfig::hash_ratio)
      .def_readwrite("prune_scale", &LatticeFasterDecoderConfig::prune_scale)
      .def_readwrite("memory_pool_tokens_block_size",
                     &LatticeFasterDecoderConfig::memory_pool_tokens_block_size)
      .def_readwrite("memory_pool_links_block_size",
                     &LatticeFasterDecoderConfig::memory_pool_links_block_size)
      .def("__str__", &LatticeFasterDecoderConfig::ToString);

  auto best_path = [](auto &self, bool use_final_probs) {
    Lattice lat;
    const bool ok = self.GetBestPath(&lat, use_final_probs);
    return std::make_pair(ok, std::move(lat));
  };
  auto raw_lattice = [](auto &self, bool use_final_probs) {
    Lattice lat;
    const bool ok = self.GetRawLattice(&lat, use_final_probs);
    return std::make_pair(ok, std::move(lat));
  };

  py::class_<LatticeSimpleDecoder>(m, "LatticeSimpleDecoder")
      .def(py::init<const Fst &, const LatticeSimpleDecoderConfig &, const DeviceOptions &>(),
           py::arg("fst"), py::arg("config"), py::arg("device_options") = DeviceOptions(),
           py::keep_alive<1, 2>())
      .def("get_config", &LatticeSimpleDecoder::GetOptions)
      .def("num_frames_decoded", &LatticeSimpleDecoder::NumFramesDecoded)
      .def("final_relative_cost", &LatticeSimpleDecoder::FinalRelativeCost)
      .def("decode", &LatticeSimpleDecoder::Decode, py::arg("decodable"))
      .def("init_decoding", &LatticeSimpleDecoder::InitDecoding)
      .def("finalize_decoding", &LatticeSimpleDecoder::FinalizeDecoding)
      .def("get_best_path",
           [best_path](LatticeSimpleDecoder &self, bool u) { return best_path(self, u); },
           py::arg("use_final_probs") = true)
      .def("get_raw_lattice",
           [raw_lattice](LatticeSimpleDecoder &self, bool u) { return raw_lattice(self, u); },
           py::arg("use_final_probs") = true);

  py::class_<LatticeFasterDecoder>(m, "LatticeFasterDecoder")
      .def(py::init<const Fst &, const LatticeFasterDecoderConfig &, const DeviceOptions &>(),
           py::arg("fst"), py::arg("config"), py::arg("device_options") = DeviceOptions(),
           py::keep_alive<1, 2>())
      .def("get_config", &LatticeFasterDecoder::GetOptions)
      .def("num_frames_decoded", &LatticeFasterDecoder::NumFramesDecoded)
      .def("final_relative_cost", &LatticeFasterDecoder::FinalRelativeCost)
      .def("decode", &LatticeFasterDecoder::Decode, py::arg("decodable"))
      .def("init_decoding", &LatticeFasterDecoder::InitDecoding)
      .def("advance_decoding", &LatticeFasterDecoder::AdvanceDecoding, py::arg("decodable"),
           py::arg("max_num_frames") = -1)
      .def("finalize_decoding", &LatticeFasterDecoder::FinalizeDecoding)
      .def("get_best_path",
           [best_path](LatticeFasterDecoder &self, bool u) { return best_path(self, u); },
           py::arg("use_final_probs") = true)
      .def("get_raw_lattice",
           [raw_lattice](LatticeFasterDecoder &self, bool u) { return raw_lattice(self, u); },
           py::arg("use_final_probs") = true);

  py::class_<BatchLatticeDecoder>(m, "BatchLatticeDecoder")
      .def(py::init<const Fst &, const LatticeFasterDecoderConfig &, int32_t, const DeviceOptions &>(),
           py::arg("fst"), py::arg("config"), py::arg("max_batch"),
           py::arg("device_options") = DeviceOptions(), py::keep_alive<1, 2>())
      .def("decode", &BatchLatticeDecoder::Decode, py::arg("decodables"))
      .def("init_decoding", &BatchLatticeDecoder::InitDecoding, py::arg("batch"))
      .def("advance_decoding", &BatchLatticeDecoder::AdvanceDecoding, py::arg("decodables"),
           py::arg("max_num_frames") = -1)
      .def("finalize_decoding", &BatchLatticeDecoder::FinalizeDecoding)
      .def("final_relative_cost", &BatchLatticeDecoder::FinalRelativeCost)
      .def("num_frames_decoded", &BatchLatticeDecoder::NumFramesDecoded)
      .def(
          "get_raw_lattice",
          [](BatchLatticeDecoder &self, bool u) {
            std::vector<Lattice> lats;
            std::vector<bool> ok = self.GetRawLattice(&lats, u);
            return std::make_pair(ok, std::move(lats));
          },
          py::arg("use_final_probs") = true)
      .def(
          "get_best_path",
          [](BatchLatticeDecoder &self, bool u) {
            std::vector<Lattice> lats;
            std::vector<bool> ok = self.GetBestPath(&lats, u);
            return std::make_pair(ok, std::move(lats));
          },
          py::arg("use_final_probs") = true)
      .def("stats", &BatchLatticeDecoder::Stats)
      .def("last_kernel_ms", &BatchLatticeDecoder::LastKernelMs);

  m.def(
      "shortest_path",
      [](const Lattice &l) {
        Lattice out;
        const bool ok = ShortestPath(l, &out);
        return std::make_pair(ok, std::move(out));
      },
      py::arg("lattice"), "tropical shortest path through a lattice (fst::ShortestPath)");

  py::class_<BatchFasterDecoder>(m, "BatchFasterDecoder")
      .def(py::init<const Fst &, const FasterDecoderOptions &, int32_t, const DeviceOptions &>(),
           py::arg("fst"), py::arg("config"), py::arg("max_batch"),
           py::arg("device_options") = DeviceOptions(), py::keep_alive<1, 2>())
      .def("set_options", &BatchFasterDecoder::SetOptions, py::arg("config"))
      .def("decode", &BatchFasterDecoder::Decode, py::arg("decodables"),
           py::call_guard<py::gil_scoped_release>())
      .def("init_decoding", &BatchFasterDecoder::InitDecoding, py::arg("batch"))
      .def("advance_decoding", &BatchFasterDecoder::AdvanceDecoding, py::arg("decodables"),
           py::arg("max_num_frames") = -1)
      .def("reached_final", &BatchFasterDecoder::ReachedFinal)
      .def("num_frames_decoded", &BatchFasterDecoder::NumFramesDecoded)
      .def(
          "get_best_path",
          [](BatchFasterDecoder &self, bool use_final_probs) {
            std::vector<Lattice> lats;
            std::vector<bool> ok = self.GetBestPath(&lats, use_final_probs);
            return std::make_pair(ok, std::move(lats));
          },
          py::arg("use_final_probs") = true)
      .def("stats", &BatchFasterDecoder::Stats)
      .def("last_kernel_ms", &BatchFasterDecoder::LastKernelMs);
}
