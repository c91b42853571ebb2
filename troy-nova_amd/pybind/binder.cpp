// binder.cpp -- `pytroy_raw`: the Python surface of the reference (pybind/src/*.cu registers the same class and method
// names) over the host-side mirror troy/troy.h.  Only what the mirror implements is exposed; everything runs on the GPU
// (call context.to_device_inplace() first), and host-resident operands raise ValueError / RuntimeError exactly as the
// C++ exceptions map (std::invalid_argument -> ValueError).
#include <pybind11/pybind11.h>
#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime_api.h>

#include <optional>
#include <sstream>

#include "../troy/conv2d.h"
#include "../troy/matmul.h"
#include "../troy/ring2k.h"
#include "../troy/troy.h"

namespace py = pybind11;
using namespace troy;

using PoolArg = std::optional<MemoryPoolHandle>;
static MemoryPoolHandle P(const PoolArg& p) { return p.has_value() ? p.value() : MemoryPool::GlobalPool(); }
#define POOL py::arg("pool") = std::nullopt
#define MODE py::arg("mode") = CompressionMode::Nil

// unary / binary / keyed evaluator operations come in (x, x_inplace, x_new) triples with identical shapes
#define EV_UNARY(cls, name)                                                                                                  \
    cls.def(#name, [](const Evaluator& s, const Ciphertext& a, Ciphertext& d, PoolArg p) { s.name(a, d, P(p)); },            \
            py::arg("encrypted"), py::arg("destination"), POOL);                                                             \
    cls.def(#name "_new", [](const Evaluator& s, const Ciphertext& a, PoolArg p) { return s.name##_new(a, P(p)); },          \
            py::arg("encrypted"), POOL)
#define EV_UNARY_INPLACE_POOL(cls, name)                                                                                     \
    cls.def(#name "_inplace", [](const Evaluator& s, Ciphertext& a, PoolArg p) { s.name##_inplace(a, P(p)); }, py::arg("encrypted"), POOL)
#define EV_BINARY(cls, name)                                                                                                 \
    cls.def(#name, [](const Evaluator& s, const Ciphertext& a, const Ciphertext& b, Ciphertext& d, PoolArg p) { s.name(a, b, d, P(p)); }, \
            py::arg("encrypted1"), py::arg("encrypted2"), py::arg("destination"), POOL);                                     \
    cls.def(#name "_inplace", [](const Evaluator& s, Ciphertext& a, const Ciphertext& b, PoolArg p) { s.name##_inplace(a, b, P(p)); },   \
            py::arg("encrypted1"), py::arg("encrypted2"), POOL);                                                             \
    cls.def(#name "_new", [](const Evaluator& s, const Ciphertext& a, const Ciphertext& b, PoolArg p) { return s.name##_new(a, b, P(p)); }, \
            py::arg("encrypted1"), py::arg("encrypted2"), POOL)
#define EV_KEYED(cls, name, KeyT, keyarg)                                                                                    \
    cls.def(#name, [](const Evaluator& s, const Ciphertext& a, const KeyT& k, Ciphertext& d, PoolArg p) { s.name(a, k, d, P(p)); }, \
            py::arg("encrypted"), py::arg(keyarg), py::arg("destination"), POOL);                                            \
    cls.def(#name "_inplace", [](const Evaluator& s, Ciphertext& a, const KeyT& k, PoolArg p) { s.name##_inplace(a, k, P(p)); },   \
            py::arg("encrypted"), py::arg(keyarg), POOL);                                                                    \
    cls.def(#name "_new", [](const Evaluator& s, const Ciphertext& a, const KeyT& k, PoolArg p) { return s.name##_new(a, k, P(p)); }, \
            py::arg("encrypted"), py::arg(keyarg), POOL)

// save(...) -> bytes / load(bytes, ...): the reference's pybind exposes the stream API through byte strings
template <typename F> static py::bytes to_bytes(F&& save) { std::ostringstream os; save(os); return py::bytes(os.str()); }

template <typename T>
static std::vector<const T*> const_ptrs(const std::vector<T*>& v) { return std::vector<const T*>(v.begin(), v.end()); }

PYBIND11_MODULE(pytroy_raw, m) {
    m.doc() = "MI355X-native troy-nova hot path: Python surface (subset of the reference's pytroy_raw)";
    m.def("it_works", []() { return 42; });
    m.def("device_count", &utils::device_count);
    m.def("initialize_kernel", [](int device) { if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("[kernel_provider::initialize] cannot select the device"); }, py::arg("device") = 0);
    m.def("destroy_memory_pool", []() { MemoryPool::Destroy(); });

    py::enum_<SchemeType>(m, "SchemeType").value("Nil", SchemeType::Nil).value("BFV", SchemeType::BFV).value("CKKS", SchemeType::CKKS).value("BGV", SchemeType::BGV);
    py::enum_<CompressionMode>(m, "CompressionMode").value("Nil", CompressionMode::Nil).value("Zstd", CompressionMode::Zstd);
    py::enum_<SecurityLevel>(m, "SecurityLevel").value("Nil", SecurityLevel::Nil).value("Classical128", SecurityLevel::Classical128)
        .value("Classical192", SecurityLevel::Classical192).value("Classical256", SecurityLevel::Classical256);

    py::class_<MemoryPool, MemoryPoolHandle>(m, "MemoryPool")
        .def(py::init([](size_t device) { return MemoryPool::create(device); }), py::arg("device") = 0)
        .def_static("global_pool", &MemoryPool::GlobalPool)
        .def_static("destroy_global_pool", &MemoryPool::Destroy).def_static("destroy", &MemoryPool::Destroy)
        .def("get_device", &MemoryPool::get_device)
        .def("release_unused", &MemoryPool::release_unused);

    py::class_<Modulus>(m, "Modulus")
        .def(py::init<uint64_t>(), py::arg("value") = 0)
        .def("value", &Modulus::value).def("bit_count", &Modulus::bit_count).def("is_prime", &Modulus::is_prime)
        .def("is_zero", &Modulus::is_zero).def("reduce", &Modulus::reduce, py::arg("input_uint64")).def("reduce_mul", &Modulus::reduce_mul_uint64, py::arg("a"), py::arg("b"))
        .def("__str__", [](const Modulus& s) { return std::to_string(s.value()); })
        .def("__repr__", [](const Modulus& s) { return "Modulus(" + std::to_string(s.value()) + ")"; });
    py::class_<CoeffModulus>(m, "CoeffModulus")
        .def_static("max_bit_count", &CoeffModulus::max_bit_count, py::arg("poly_modulus_degree"), py::arg("sec_level") = SecurityLevel::Classical128)
        .def_static("create", &CoeffModulus::create_vector, py::arg("poly_modulus_degree"), py::arg("bit_sizes"))
        .def_static("bfv_default", [](size_t n, SecurityLevel sec) { return CoeffModulus::bfv_default_vector(n, sec); }, py::arg("poly_modulus_degree"), py::arg("sec_level") = SecurityLevel::Classical128);
    py::class_<PlainModulus>(m, "PlainModulus").def_static("batching", &PlainModulus::batching, py::arg("poly_modulus_degree"), py::arg("bit_size"))
        .def_static("batching_multiple", [](size_t n, const std::vector<size_t>& bits) { return PlainModulus::batching_multiple(n, bits).to_vector(); }, py::arg("poly_modulus_degree"), py::arg("bit_sizes"));

    py::class_<ParmsID>(m, "ParmsID")
        .def("is_zero", &ParmsID::is_zero).def_static("zero", []() { return parms_id_zero; })
        .def("__eq__", [](const ParmsID& a, const ParmsID& b) { return a == b; })
        .def("__hash__", [](const ParmsID& a) { return ParmsIDHash{}(a); })
        .def("to_vector", [](const ParmsID& a) { return py::array_t<uint64_t>(4, a.v); });
    m.attr("parms_id_zero") = parms_id_zero;

    py::class_<EncryptionParameters>(m, "EncryptionParameters")
        .def(py::init<SchemeType>(), py::arg("scheme"))
        .def("set_poly_modulus_degree", &EncryptionParameters::set_poly_modulus_degree)
        .def("set_coeff_modulus", [](EncryptionParameters& s, const std::vector<Modulus>& q) { s.set_coeff_modulus(q); })
        .def("set_coeff_modulus", [](EncryptionParameters& s, const std::vector<uint64_t>& q) { s.set_coeff_modulus(q); })
        .def("set_plain_modulus", py::overload_cast<const Modulus&>(&EncryptionParameters::set_plain_modulus))
        .def("set_plain_modulus", py::overload_cast<uint64_t>(&EncryptionParameters::set_plain_modulus))
        .def("scheme", &EncryptionParameters::scheme).def("poly_modulus_degree", &EncryptionParameters::poly_modulus_degree)
        .def("coeff_modulus", [](const EncryptionParameters& s) { return s.coeff_modulus().to_vector(); }).def("plain_modulus", [](const EncryptionParameters& s) { return Modulus(s.plain_modulus().value()); })
        .def("parms_id", &EncryptionParameters::parms_id)
        .def("pool", [](const EncryptionParameters&) { return MemoryPool::GlobalPool(); }).def("device_index", [](const EncryptionParameters&) { return size_t(0); })
        .def("save", [](const EncryptionParameters& s, CompressionMode mode) { if (mode != CompressionMode::Nil) throw std::invalid_argument("[serialize::compress] Zstd is not available in this build."); return to_bytes([&](std::ostream& os) { s.save(os); }); }, MODE)
        .def("load", [](EncryptionParameters& s, const py::bytes& b) { std::istringstream is{std::string(b)}; s.load(is); }, py::arg("str"))
        .def_static("load_new", [](const py::bytes& b) { std::istringstream is{std::string(b)}; EncryptionParameters e(SchemeType::Nil); e.load(is); return e; }, py::arg("str"))
        .def("serialized_size_upperbound", [](const EncryptionParameters& s, CompressionMode) { std::ostringstream os; return s.save(os); }, MODE);

    using CDP = std::shared_ptr<ContextData>;
    auto unconst = [](const std::optional<ContextDataPointer>& c) -> std::optional<CDP> {
        return c.has_value() ? std::optional<CDP>(std::const_pointer_cast<ContextData>(c.value())) : std::nullopt;
    };
    py::class_<ContextData, CDP>(m, "ContextData")
        .def("parms", &ContextData::parms).def("parms_id", &ContextData::parms_id).def("chain_index", &ContextData::chain_index)
        .def("device_index", [](const ContextData&) { return size_t(0); })
        .def("next_context_data", [unconst](const ContextData& s) { return unconst(s.next_context_data()); })
        .def("prev_context_data", [unconst](const ContextData& s) { auto p = s.prev_context_data_pointer().lock(); return unconst(p ? std::optional<ContextDataPointer>(p) : std::nullopt); });

    py::class_<HeContext, HeContextPointer>(m, "HeContext")
        .def(py::init([](const EncryptionParameters& parms, bool expand, SecurityLevel sec, uint64_t seed) { return HeContext::create(parms, expand, sec, seed); }),
             py::arg("parms"), py::arg("expand_mod_chain") = true, py::arg("sec_level") = SecurityLevel::Classical128, py::arg("random_seed") = 0)
        .def("to_device_inplace", [](HeContext& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL)
        .def("on_device", &HeContext::on_device).def("pool", &HeContext::pool).def("device_index", &HeContext::device_index)
        .def("key_parms_id", &HeContext::key_parms_id).def("first_parms_id", &HeContext::first_parms_id).def("last_parms_id", &HeContext::last_parms_id)
        .def("get_context_data", [unconst](const HeContext& s, const ParmsID& id) { return unconst(s.get_context_data(id)); })
        .def("key_context_data", [unconst](const HeContext& s) { return unconst(s.key_context_data()); })
        .def("first_context_data", [unconst](const HeContext& s) { return unconst(s.first_context_data()); })
        .def("last_context_data", [unconst](const HeContext& s) { return unconst(s.last_context_data()); })
        .def("using_keyswitching", &HeContext::using_keyswitching).def("parameters_set", &HeContext::parameters_set);

    py::class_<Plaintext>(m, "Plaintext")
        .def(py::init<>())
        .def("clone", [](const Plaintext& s, PoolArg p) { return s.clone(P(p)); }, POOL)
        .def("on_device", &Plaintext::on_device)
        .def("to_device_inplace", [](Plaintext& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL)
        .def("to_host_inplace", &Plaintext::to_host_inplace)
        .def("parms_id", [](const Plaintext& s) { return s.parms_id(); }).def("scale", [](const Plaintext& s) { return s.scale(); })
        .def("set_scale", [](Plaintext& s, double v) { s.scale() = v; })
        .def("coeff_count", [](const Plaintext& s) { return s.coeff_count(); }).def("is_ntt_form", [](const Plaintext& s) { return s.is_ntt_form(); })
        .def("data", [](const Plaintext& s) { return s.data().to_vector(); })
        .def("obtain_data", [](const Plaintext& s) { const std::vector<uint64_t> v = s.data().to_vector(); return py::array_t<uint64_t>(v.size(), v.data()); })
        .def("address", [](const Plaintext& s) { return reinterpret_cast<uintptr_t>(&s); })
        .def("data_address", [](const Plaintext& s) { return reinterpret_cast<uintptr_t>(s.data().raw_pointer()); })
        .def("pool", &Plaintext::pool).def("device_index", [](const Plaintext& s) { return s.pool() ? s.pool()->get_device() : size_t(0); })
        .def("to_device", [](const Plaintext& s, PoolArg p) { return s.to_device(P(p)); }, POOL).def("to_host", &Plaintext::to_host)
        .def("set_parms_id", [](Plaintext& s, const ParmsID& id) { s.parms_id() = id; }).def("set_coeff_count", [](Plaintext& s, size_t c) { s.coeff_count() = c; })
        .def("set_is_ntt_form", [](Plaintext& s, bool f) { s.is_ntt_form() = f; }).def("resize", &Plaintext::resize, py::arg("coeff_count"), py::arg("fill_extra_with_zeros") = true, py::arg("copy_data") = true)
        .def("coeff_modulus_size", [](const Plaintext& s) { return s.coeff_modulus_size(); }).def("poly_modulus_degree", [](const Plaintext& s) { return s.poly_modulus_degree(); })
        .def("serialized_size_upperbound", [](const Plaintext& s, CompressionMode mode) { return s.serialized_size_upperbound(mode); }, MODE)
        .def("save", [](const Plaintext& s, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.save(os, mode); }); }, MODE)
        .def("load", [](Plaintext& s, const std::string& b, PoolArg p) { std::istringstream is(b); s.load(is, P(p)); }, py::arg("str"), POOL)
        .def_static("load_new", [](const std::string& b, PoolArg p) { std::istringstream is(b); return Plaintext::load_new(is, P(p)); }, py::arg("str"), POOL);

    py::class_<Ciphertext>(m, "Ciphertext")
        .def(py::init<>())
        .def("clone", [](const Ciphertext& s, PoolArg p) { return s.clone(P(p)); }, POOL)
        .def("on_device", &Ciphertext::on_device)
        .def("to_device_inplace", [](Ciphertext& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL)
        .def("to_host_inplace", &Ciphertext::to_host_inplace)
        .def("parms_id", [](const Ciphertext& s) { return s.parms_id(); }).def("scale", [](const Ciphertext& s) { return s.scale(); })
        .def("set_scale", [](Ciphertext& s, double v) { s.scale() = v; })
        .def("polynomial_count", &Ciphertext::polynomial_count).def("coeff_modulus_size", &Ciphertext::coeff_modulus_size)
        .def("poly_modulus_degree", &Ciphertext::poly_modulus_degree)
        .def("is_ntt_form", [](const Ciphertext& s) { return s.is_ntt_form(); }).def("contains_seed", &Ciphertext::contains_seed)
        .def("data", [](const Ciphertext& s) { return s.data().to_vector(); })
        .def("expand_seed", &Ciphertext::expand_seed)
        .def("obtain_data", [](const Ciphertext& s) { const std::vector<uint64_t> v = s.data().to_vector(); return py::array_t<uint64_t>(v.size(), v.data()); })
        .def("address", [](const Ciphertext& s) { return reinterpret_cast<uintptr_t>(&s); })
        .def("data_address", [](const Ciphertext& s) { return reinterpret_cast<uintptr_t>(s.data().raw_pointer()); })
        .def("pool", &Ciphertext::pool).def("device_index", [](const Ciphertext& s) { return s.pool() ? s.pool()->get_device() : size_t(0); })
        .def("to_device", [](const Ciphertext& s, PoolArg p) { return s.to_device(P(p)); }, POOL).def("to_host", &Ciphertext::to_host)
        .def("set_parms_id", [](Ciphertext& s, const ParmsID& id) { s.parms_id() = id; }).def("set_is_ntt_form", [](Ciphertext& s, bool f) { s.is_ntt_form() = f; })
        .def("correction_factor", [](const Ciphertext& s) { return s.correction_factor(); }).def("set_correction_factor", [](Ciphertext& s, uint64_t f) { s.correction_factor() = f; })
        .def("seed", [](const Ciphertext& s) { return s.seed(); }).def("set_seed", [](Ciphertext& s, uint64_t v) { s.seed() = v; })
        .def("is_transparent", &Ciphertext::is_transparent)
        .def("serialized_size_upperbound", [](const Ciphertext& s, HeContextPointer c, CompressionMode mode) { return s.serialized_size_upperbound(c, mode); }, py::arg("context"), MODE)
        .def("save_terms", [](const Ciphertext& s, HeContextPointer c, const std::vector<size_t>& terms, PoolArg p, CompressionMode mode) {
            return to_bytes([&](std::ostream& os) { s.save_terms(os, c, terms, P(p), mode); }); }, py::arg("context"), py::arg("terms"), POOL, MODE)
        .def("load_terms", [](Ciphertext& s, const py::bytes& b, HeContextPointer c, const std::vector<size_t>& terms, PoolArg p) {
            std::istringstream is{std::string(b)}; s.load_terms(is, c, terms, P(p)); }, py::arg("str"), py::arg("context"), py::arg("terms"), POOL)
        .def_static("load_terms_new", [](const py::bytes& b, HeContextPointer c, const std::vector<size_t>& terms, PoolArg p) {
            std::istringstream is{std::string(b)}; return Ciphertext::load_terms_new(is, c, terms, P(p)); }, py::arg("str"), py::arg("context"), py::arg("terms"), POOL)
        .def("serialized_terms_size_upperbound", [](const Ciphertext& s, HeContextPointer c, const std::vector<size_t>& terms, CompressionMode mode) {
            return s.serialized_terms_size_upperbound(c, terms.size(), mode); }, py::arg("context"), py::arg("terms"), MODE)
        .def("save", [](const Ciphertext& s, HeContextPointer c, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.save(os, c, mode); }); }, py::arg("context"), MODE)
        .def("load", [](Ciphertext& s, const std::string& b, HeContextPointer c, PoolArg p) { std::istringstream is(b); s.load(is, c, P(p)); }, py::arg("str"), py::arg("context"), POOL)
        .def_static("load_new", [](const std::string& b, HeContextPointer c, PoolArg p) { std::istringstream is(b); return Ciphertext::load_new(is, c, P(p)); },
                    py::arg("str"), py::arg("context"), POOL);

    py::class_<SecretKey>(m, "SecretKey").def(py::init<>()).def(py::init([](const Plaintext& p) { return SecretKey(Plaintext(p)); })).def("on_device", &SecretKey::on_device)
        .def("to_device_inplace", [](SecretKey& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL).def("to_host_inplace", &SecretKey::to_host_inplace)
        .def("to_device", [](const SecretKey& s, PoolArg p) { return s.to_device(P(p)); }, POOL).def("to_host", &SecretKey::to_host)
        .def("parms_id", [](const SecretKey& s) { return s.parms_id(); }).def("set_parms_id", [](SecretKey& s, const ParmsID& id) { s.parms_id() = id; })
        .def("pool", [](const SecretKey& s) { return s.as_plaintext().pool(); }).def("device_index", [](const SecretKey& s) { return s.as_plaintext().pool() ? s.as_plaintext().pool()->get_device() : size_t(0); })
        .def("as_plaintext", [](const SecretKey& s) { return s.as_plaintext(); }).def("get_plaintext", [](const SecretKey& s, PoolArg p) { return s.as_plaintext().clone(P(p)); }, POOL)
        .def("load", [](SecretKey& s, const py::bytes& b, PoolArg p) { std::istringstream is{std::string(b)}; s.load(is, P(p)); }, py::arg("str"), POOL)
        .def("serialized_size_upperbound", [](const SecretKey& s, CompressionMode mode) { return s.serialized_size_upperbound(mode); }, MODE)
        .def("clone", [](const SecretKey& s, PoolArg p) { return s.clone(P(p)); }, POOL).def("data", [](const SecretKey& s) { return s.data().to_vector(); })
        .def("save", [](const SecretKey& s, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.save(os, mode); }); }, MODE)
        .def_static("load_new", [](const std::string& b, PoolArg p) { std::istringstream is(b); return SecretKey::load_new(is, P(p)); }, py::arg("str"), POOL);
    py::class_<PublicKey>(m, "PublicKey").def(py::init<>()).def("on_device", &PublicKey::on_device)
        .def("to_device_inplace", [](PublicKey& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL).def("to_host_inplace", &PublicKey::to_host_inplace)
        .def("to_device", [](const PublicKey& s, PoolArg p) { return s.to_device(P(p)); }, POOL).def("to_host", &PublicKey::to_host)
        .def("parms_id", [](const PublicKey& s) { return s.parms_id(); }).def("set_parms_id", [](PublicKey& s, const ParmsID& id) { s.parms_id() = id; })
        .def("pool", [](const PublicKey& s) { return s.as_ciphertext().pool(); }).def("device_index", [](const PublicKey& s) { return s.as_ciphertext().pool() ? s.as_ciphertext().pool()->get_device() : size_t(0); })
        .def("get_ciphertext", [](const PublicKey& s, PoolArg p) { return s.as_ciphertext().clone(P(p)); }, POOL)
        .def("contains_seed", &PublicKey::contains_seed).def("expand_seed", &PublicKey::expand_seed)
        .def("save", [](const PublicKey& s, HeContextPointer c, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.save(os, c, mode); }); }, py::arg("context"), MODE)
        .def("load", [](PublicKey& s, const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; s.load(is, c, P(p)); }, py::arg("str"), py::arg("context"), POOL)
        .def_static("load_new", [](const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; return PublicKey::load_new(is, c, P(p)); },
                    py::arg("str"), py::arg("context"), POOL)
        .def("serialized_size_upperbound", [](const PublicKey& s, HeContextPointer c, CompressionMode mode) { return s.serialized_size_upperbound(c, mode); }, py::arg("context"), MODE)
        .def("clone", [](const PublicKey& s, PoolArg p) { return s.clone(P(p)); }, POOL).def("as_ciphertext", [](const PublicKey& s) { return s.as_ciphertext(); });
    py::class_<KSwitchKeys>(m, "KSwitchKeys").def(py::init<>()).def("on_device", &KSwitchKeys::on_device).def("parms_id", [](const KSwitchKeys& s) { return s.parms_id(); })
        .def("set_parms_id", [](KSwitchKeys& s, const ParmsID& id) { s.parms_id() = id; })
        .def("pool", &KSwitchKeys::pool).def("device_index", [](const KSwitchKeys& s) { return s.pool() ? s.pool()->get_device() : size_t(0); })
        .def("serialized_size_upperbound", [](const KSwitchKeys& s, HeContextPointer c, CompressionMode mode) { return s.serialized_size_upperbound(c, mode); }, py::arg("context"), MODE)
        .def("clone", [](const KSwitchKeys& s, PoolArg p) { return s.clone(P(p)); }, POOL)
        .def("to_device_inplace", [](KSwitchKeys& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL).def("to_host_inplace", &KSwitchKeys::to_host_inplace)
        .def("to_device", [](const KSwitchKeys& s, PoolArg p) { return s.to_device(P(p)); }, POOL).def("to_host", &KSwitchKeys::to_host)
        .def("as_kswitch_keys", [](const KSwitchKeys& s) { return s.as_kswitch_keys(); }).def("get_kswitch_keys", [](const KSwitchKeys& s, PoolArg p) { return s.as_kswitch_keys().clone(P(p)); }, POOL)
        .def_static("load_new", [](const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; KSwitchKeys k; k.load(is, c, P(p)); return k; },
                    py::arg("str"), py::arg("context"), POOL)
        .def("save", [](const KSwitchKeys& s, HeContextPointer c, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.save(os, c, mode); }); }, py::arg("context"), MODE)
        .def("load", [](KSwitchKeys& s, const std::string& b, HeContextPointer c, PoolArg p) { std::istringstream is(b); s.load(is, c, P(p)); }, py::arg("str"), py::arg("context"), POOL);
    py::class_<RelinKeys, KSwitchKeys>(m, "RelinKeys").def(py::init<>()).def("has_key", &RelinKeys::has_key)
        // the reference registers RelinKeys as a class of its own: load_new / clone / to_device / to_host return RelinKeys, not KSwitchKeys
        .def_static("load_new", [](const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; KSwitchKeys k; k.load(is, c, P(p)); return RelinKeys(std::move(k)); },
                    py::arg("str"), py::arg("context"), POOL)
        .def("clone", [](const RelinKeys& s, PoolArg p) { (void)p; return RelinKeys(s); }, POOL)
        .def("to_device", [](const RelinKeys& s, PoolArg p) { RelinKeys k(s); k.to_device_inplace(P(p)); return k; }, POOL)
        .def("to_host", [](const RelinKeys& s) { RelinKeys k(s); k.to_host_inplace(); return k; });
    py::class_<GaloisKeys, KSwitchKeys>(m, "GaloisKeys").def(py::init<>()).def("has_key", &GaloisKeys::has_key)
        // the reference registers GaloisKeys as a class of its own: load_new / clone / to_device / to_host return GaloisKeys, not KSwitchKeys
        .def_static("load_new", [](const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; KSwitchKeys k; k.load(is, c, P(p)); return GaloisKeys(std::move(k)); },
                    py::arg("str"), py::arg("context"), POOL)
        .def("clone", [](const GaloisKeys& s, PoolArg p) { (void)p; return GaloisKeys(s); }, POOL)
        .def("to_device", [](const GaloisKeys& s, PoolArg p) { GaloisKeys k(s); k.to_device_inplace(P(p)); return k; }, POOL)
        .def("to_host", [](const GaloisKeys& s) { GaloisKeys k(s); k.to_host_inplace(); return k; });

    py::class_<KeyGenerator>(m, "KeyGenerator")
        .def(py::init([](HeContextPointer c, PoolArg p) { return new KeyGenerator(c, P(p)); }), py::arg("context"), POOL)
        .def(py::init([](HeContextPointer c, const SecretKey& sk, PoolArg p) { return new KeyGenerator(c, sk, P(p)); }), py::arg("context"), py::arg("secret_key"), POOL)
        .def("on_device", &KeyGenerator::on_device).def("context", &KeyGenerator::context).def("to_device_inplace", [](KeyGenerator&, PoolArg) {}, POOL)
        .def("secret_key", [](const KeyGenerator& s) { return s.secret_key().clone(); })
        .def("create_public_key", [](const KeyGenerator& s, bool save_seed, PoolArg p) { return s.create_public_key(save_seed, P(p)); }, py::arg("save_seed"), POOL)
        .def("create_relin_keys", [](const KeyGenerator& s, bool save_seed, size_t max_power, PoolArg p) { return s.create_relin_keys(save_seed, max_power, P(p)); },
             py::arg("save_seed"), py::arg("max_power") = 2, POOL)
        .def("create_keyswitching_key", [](const KeyGenerator& s, const SecretKey& nk, bool save_seed, PoolArg p) { return s.create_keyswitching_key(nk, save_seed, P(p)); },
             py::arg("secret_key"), py::arg("save_seed"), POOL)
        .def("create_galois_keys", [](const KeyGenerator& s, bool save_seed, PoolArg p) { return s.create_galois_keys(save_seed, P(p)); }, py::arg("save_seed"), POOL)
        .def("create_automorphism_keys", [](const KeyGenerator& s, bool save_seed, PoolArg p) { return s.create_automorphism_keys(save_seed, P(p)); }, py::arg("save_seed"), POOL)
        .def("create_galois_keys_from_steps", [](const KeyGenerator& s, const std::vector<int>& st, bool save_seed, PoolArg p) { return s.create_galois_keys_from_steps(st, save_seed, P(p)); },
             py::arg("galois_steps"), py::arg("save_seed"), POOL)
        .def("create_galois_keys_from_elements", [](const KeyGenerator& s, const std::vector<size_t>& el, bool save_seed, PoolArg p) { return s.create_galois_keys_from_elements(el, save_seed, P(p)); },
             py::arg("galois_elts"), py::arg("save_seed"), POOL);

    py::class_<utils::RandomGenerator>(m, "RandomGenerator")
        .def(py::init([]() { return new utils::RandomGenerator(); })).def(py::init([](uint64_t seed) { return new utils::RandomGenerator(seed); }), py::arg("seed"))
        .def("reset_seed", [](utils::RandomGenerator& s, uint64_t seed) { s.reset_seed(seed); }, py::arg("seed"))
        .def("sample_uint64", &utils::RandomGenerator::sample_uint64);

    py::class_<Encryptor>(m, "Encryptor")
        .def(py::init<HeContextPointer>()).def("context", &Encryptor::context)
        .def("to_device_inplace", [](Encryptor&, PoolArg) {}, POOL)
        .def("set_public_key", [](Encryptor& s, const PublicKey& k, PoolArg p) { s.set_public_key(k, P(p)); }, py::arg("public_key"), POOL)
        .def("set_secret_key", [](Encryptor& s, const SecretKey& k, PoolArg p) { s.set_secret_key(k, P(p)); }, py::arg("secret_key"), POOL)
        .def("encrypt_asymmetric", [](const Encryptor& s, const Plaintext& pl, Ciphertext& d, PoolArg p) { s.encrypt_asymmetric(pl, d, P(p)); }, py::arg("plain"), py::arg("destination"), POOL)
        .def("encrypt_asymmetric_new", [](const Encryptor& s, const Plaintext& pl, PoolArg p) { return s.encrypt_asymmetric_new(pl, P(p)); }, py::arg("plain"), POOL)
        .def("encrypt_symmetric", [](const Encryptor& s, const Plaintext& pl, bool seed, Ciphertext& d, PoolArg p) { s.encrypt_symmetric(pl, seed, d, P(p)); },
             py::arg("plain"), py::arg("save_seed"), py::arg("destination"), POOL)
        .def("encrypt_symmetric_new", [](const Encryptor& s, const Plaintext& pl, bool seed, PoolArg p) { return s.encrypt_symmetric_new(pl, seed, P(p)); },
             py::arg("plain"), py::arg("save_seed"), POOL)
        .def("on_device", [](const Encryptor& s) { return s.context()->on_device(); })
        // the overloads with a caller-supplied generator (pybind/src/encryptor.cu: py::arg("rng") before the pool)
        .def("encrypt_asymmetric", [](const Encryptor& s, const Plaintext& pl, Ciphertext& d, utils::RandomGenerator& rng, PoolArg p) { s.encrypt_asymmetric(pl, d, &rng, P(p)); },
             py::arg("plain"), py::arg("destination"), py::arg("rng"), POOL)
        .def("encrypt_asymmetric_new", [](const Encryptor& s, const Plaintext& pl, utils::RandomGenerator& rng, PoolArg p) { return s.encrypt_asymmetric_new(pl, &rng, P(p)); },
             py::arg("plain"), py::arg("rng"), POOL)
        .def("encrypt_symmetric", [](const Encryptor& s, const Plaintext& pl, bool seed, Ciphertext& d, utils::RandomGenerator& rng, PoolArg p) { s.encrypt_symmetric(pl, seed, d, &rng, P(p)); },
             py::arg("plain"), py::arg("save_seed"), py::arg("destination"), py::arg("rng"), POOL)
        .def("encrypt_symmetric_new", [](const Encryptor& s, const Plaintext& pl, bool seed, utils::RandomGenerator& rng, PoolArg p) { return s.encrypt_symmetric_new(pl, seed, &rng, P(p)); },
             py::arg("plain"), py::arg("save_seed"), py::arg("rng"), POOL)
        .def("encrypt_zero_asymmetric", [](const Encryptor& s, Ciphertext& d, std::optional<ParmsID> id, utils::RandomGenerator& rng, PoolArg p) { s.encrypt_zero_asymmetric(d, id, &rng, P(p)); },
             py::arg("destination"), py::arg("parms_id"), py::arg("rng"), POOL)
        .def("encrypt_zero_asymmetric_new", [](const Encryptor& s, std::optional<ParmsID> id, utils::RandomGenerator& rng, PoolArg p) { return s.encrypt_zero_asymmetric_new(id, &rng, P(p)); },
             py::arg("parms_id"), py::arg("rng"), POOL)
        .def("encrypt_zero_symmetric", [](const Encryptor& s, bool seed, Ciphertext& d, std::optional<ParmsID> id, utils::RandomGenerator& rng, PoolArg p) { s.encrypt_zero_symmetric(seed, d, id, &rng, P(p)); },
             py::arg("save_seed"), py::arg("destination"), py::arg("parms_id"), py::arg("rng"), POOL)
        .def("encrypt_zero_symmetric_new", [](const Encryptor& s, bool seed, std::optional<ParmsID> id, utils::RandomGenerator& rng, PoolArg p) { return s.encrypt_zero_symmetric_new(seed, id, &rng, P(p)); },
             py::arg("save_seed"), py::arg("parms_id"), py::arg("rng"), POOL)
        .def("public_key", [](const Encryptor& s) { return s.public_key(); }).def("secret_key", [](const Encryptor& s) { return s.secret_key(); })
        .def("encrypt_zero_asymmetric", [](const Encryptor& s, Ciphertext& d, std::optional<ParmsID> id, PoolArg p) { d = s.encrypt_zero_asymmetric_new(id, P(p)); },
             py::arg("destination"), py::arg("parms_id") = std::nullopt, POOL)
        .def("encrypt_zero_symmetric", [](const Encryptor& s, bool seed, Ciphertext& d, std::optional<ParmsID> id, PoolArg p) { d = s.encrypt_zero_symmetric_new(seed, id, P(p)); },
             py::arg("save_seed"), py::arg("destination"), py::arg("parms_id") = std::nullopt, POOL)
        .def("encrypt_zero_asymmetric_new", [](const Encryptor& s, std::optional<ParmsID> id, PoolArg p) { return s.encrypt_zero_asymmetric_new(id, P(p)); },
             py::arg("parms_id") = std::nullopt, POOL)
        .def("encrypt_zero_symmetric_new", [](const Encryptor& s, bool seed, std::optional<ParmsID> id, PoolArg p) { return s.encrypt_zero_symmetric_new(seed, id, P(p)); },
             py::arg("save_seed"), py::arg("parms_id") = std::nullopt, POOL);

    py::class_<Decryptor>(m, "Decryptor")
        .def(py::init([](HeContextPointer c, const SecretKey& sk, PoolArg p) { return new Decryptor(c, sk, P(p)); }), py::arg("context"), py::arg("secret_key"), POOL)
        .def("to_device_inplace", [](Decryptor&, PoolArg) {}, POOL).def("on_device", &Decryptor::on_device)
        .def("decrypt", [](const Decryptor& s, const Ciphertext& c, Plaintext& d, PoolArg p) { s.decrypt(c, d, P(p)); }, py::arg("encrypted"), py::arg("destination"), POOL)
        .def("invariant_noise_budget", [](const Decryptor& s, const Ciphertext& c, PoolArg p) { return s.invariant_noise_budget(c, P(p)); }, py::arg("encrypted"), POOL)
        .def("decrypt_new", [](const Decryptor& s, const Ciphertext& c, PoolArg p) { return s.decrypt_new(c, P(p)); }, py::arg("encrypted"), POOL)
        .def("context", &Decryptor::context)
        .def("bfv_decrypt_without_scaling_down", [](const Decryptor& s, const Ciphertext& c, Plaintext& d, PoolArg p) { s.bfv_decrypt_without_scaling_down(c, d, P(p)); },
             py::arg("encrypted"), py::arg("destination"), POOL)
        .def("bfv_decrypt_without_scaling_down_new", [](const Decryptor& s, const Ciphertext& c, PoolArg p) { return s.bfv_decrypt_without_scaling_down_new(c, P(p)); }, py::arg("encrypted"), POOL);

    py::class_<BatchEncoder>(m, "BatchEncoder")
        .def(py::init<HeContextPointer>()).def("context", &BatchEncoder::context).def("slot_count", &BatchEncoder::slot_count)
        .def("on_device", &BatchEncoder::on_device).def("to_device_inplace", [](BatchEncoder&, PoolArg) {}, POOL)
        .def("encode_simd", [](const BatchEncoder& s, const std::vector<uint64_t>& v, Plaintext& d, PoolArg p) { s.encode(v, d, P(p)); }, py::arg("values"), py::arg("destination"), POOL)
        .def("encode_simd_new", [](const BatchEncoder& s, const std::vector<uint64_t>& v, PoolArg p) { return s.encode_new(v, P(p)); }, py::arg("values"), POOL)
        .def("decode_simd_new", [](const BatchEncoder& s, const Plaintext& pl, PoolArg p) { const std::vector<uint64_t> v = s.decode_new(pl, P(p)); return py::array_t<uint64_t>(v.size(), v.data()); }, py::arg("source"), POOL)
        .def("encode_polynomial_new", [](const BatchEncoder& s, const std::vector<uint64_t>& v, PoolArg p) { return s.encode_polynomial_new(v, P(p)); }, py::arg("values"), POOL)
        .def("decode_polynomial_new", [](const BatchEncoder& s, const Plaintext& pl, PoolArg p) { const std::vector<uint64_t> v = s.decode_polynomial_new(pl, P(p)); return py::array_t<uint64_t>(v.size(), v.data()); }, py::arg("source"), POOL)
        .def("encode_polynomial", [](const BatchEncoder& s, const std::vector<uint64_t>& v, Plaintext& d, PoolArg p) { s.encode_polynomial(v, d, P(p)); }, py::arg("values"), py::arg("destination"), POOL)
        .def("row_count", &BatchEncoder::row_count).def("column_count", &BatchEncoder::column_count).def("simd_encoding_supported", &BatchEncoder::simd_encoding_supported)
        .def("scale_up", [](const BatchEncoder& s, const Plaintext& a, Plaintext& d, std::optional<ParmsID> id, PoolArg p) { s.scale_up(a, d, id, P(p)); },
             py::arg("source"), py::arg("destination"), py::arg("parms_id") = std::nullopt, POOL)
        .def("scale_up_inplace", [](const BatchEncoder& s, Plaintext& a, std::optional<ParmsID> id, PoolArg p) { s.scale_up_inplace(a, id, P(p)); }, py::arg("source"), py::arg("parms_id") = std::nullopt, POOL)
        .def("scale_up_new", [](const BatchEncoder& s, const Plaintext& a, std::optional<ParmsID> id, PoolArg p) { return s.scale_up_new(a, id, P(p)); }, py::arg("source"), py::arg("parms_id") = std::nullopt, POOL)
        .def("centralize", [](const BatchEncoder& s, const Plaintext& a, Plaintext& d, std::optional<ParmsID> id, PoolArg p) { s.centralize(a, d, id, P(p)); },
             py::arg("source"), py::arg("destination"), py::arg("parms_id") = std::nullopt, POOL)
        .def("centralize_inplace", [](const BatchEncoder& s, Plaintext& a, std::optional<ParmsID> id, PoolArg p) { s.centralize_inplace(a, id, P(p)); }, py::arg("source"), py::arg("parms_id") = std::nullopt, POOL)
        .def("centralize_new", [](const BatchEncoder& s, const Plaintext& a, std::optional<ParmsID> id, PoolArg p) { return s.centralize_new(a, id, P(p)); }, py::arg("source"), py::arg("parms_id") = std::nullopt, POOL)
        .def("scale_down", [](const BatchEncoder& s, const Plaintext& a, Plaintext& d, PoolArg p) { s.scale_down(a, d, P(p)); }, py::arg("source"), py::arg("destination"), POOL)
        .def("scale_down_inplace", [](const BatchEncoder& s, Plaintext& a, PoolArg p) { s.scale_down_inplace(a, P(p)); }, py::arg("source"), POOL)
        .def("scale_down_new", [](const BatchEncoder& s, const Plaintext& a, PoolArg p) { return s.scale_down_new(a, P(p)); }, py::arg("source"), POOL)
        .def("decentralize_new", [](const BatchEncoder& s, const Plaintext& a, uint64_t cf, PoolArg p) { return s.decentralize_new(a, cf, P(p)); }, py::arg("source"), py::arg("correction_factor") = 1, POOL);

    py::class_<CKKSEncoder>(m, "CKKSEncoder")
        .def(py::init<HeContextPointer>()).def("context", &CKKSEncoder::context).def("slot_count", &CKKSEncoder::slot_count)
        .def("polynomial_modulus_degree", &CKKSEncoder::polynomial_modulus_degree).def("on_device", &CKKSEncoder::on_device)
        .def("to_device_inplace", [](CKKSEncoder&, PoolArg) {}, POOL)
        .def("encode_complex64_simd_new", [](const CKKSEncoder& s, const std::vector<std::complex<double>>& v, std::optional<ParmsID> id, double scale, PoolArg p) {
            return s.encode_complex64_simd_new(v, id, scale, P(p)); }, py::arg("values"), py::arg("parms_id"), py::arg("scale"), POOL)
        .def("encode_float64_polynomial_new", [](const CKKSEncoder& s, const std::vector<double>& v, std::optional<ParmsID> id, double scale, PoolArg p) {
            return s.encode_float64_polynomial_new(v, id, scale, P(p)); }, py::arg("values"), py::arg("parms_id"), py::arg("scale"), POOL)
        .def("encode_float64_single_new", [](const CKKSEncoder& s, double v, std::optional<ParmsID> id, double scale, PoolArg p) {
            return s.encode_float64_single_new(v, id, scale, P(p)); }, py::arg("value"), py::arg("parms_id"), py::arg("scale"), POOL)
        .def("decode_complex64_simd_new", [](const CKKSEncoder& s, const Plaintext& pl, PoolArg p) { const std::vector<std::complex<double>> v = s.decode_complex64_simd_new(pl, P(p)); return py::array_t<std::complex<double>>(v.size(), v.data()); }, py::arg("source"), POOL)
        .def("decode_float64_polynomial_new", [](const CKKSEncoder& s, const Plaintext& pl, PoolArg p) { const std::vector<double> v = s.decode_float64_polynomial_new(pl, P(p)); return py::array_t<double>(v.size(), v.data()); }, py::arg("source"), POOL)
        .def("poly_modulus_degree", &CKKSEncoder::polynomial_modulus_degree)
        .def("encode_complex64_simd", [](const CKKSEncoder& s, const std::vector<std::complex<double>>& v, std::optional<ParmsID> id, double scale, Plaintext& d, PoolArg p) {
            s.encode_complex64_simd(v, id, scale, d, P(p)); }, py::arg("values"), py::arg("parms_id"), py::arg("scale"), py::arg("destination"), POOL)
        .def("encode_float64_polynomial", [](const CKKSEncoder& s, const std::vector<double>& v, std::optional<ParmsID> id, double scale, Plaintext& d, PoolArg p) {
            s.encode_float64_polynomial(v, id, scale, d, P(p)); }, py::arg("values"), py::arg("parms_id"), py::arg("scale"), py::arg("destination"), POOL)
        .def("encode_float64_single", [](const CKKSEncoder& s, double v, std::optional<ParmsID> id, double scale, Plaintext& d, PoolArg p) {
            s.encode_float64_single(v, id, scale, d, P(p)); }, py::arg("value"), py::arg("parms_id"), py::arg("scale"), py::arg("destination"), POOL)
        .def("encode_complex64_single", [](const CKKSEncoder& s, std::complex<double> v, std::optional<ParmsID> id, double scale, Plaintext& d, PoolArg p) {
            s.encode_complex64_single(v, id, scale, d, P(p)); }, py::arg("value"), py::arg("parms_id"), py::arg("scale"), py::arg("destination"), POOL)
        .def("encode_complex64_single_new", [](const CKKSEncoder& s, std::complex<double> v, std::optional<ParmsID> id, double scale, PoolArg p) {
            return s.encode_complex64_single_new(v, id, scale, P(p)); }, py::arg("value"), py::arg("parms_id"), py::arg("scale"), POOL)
        .def("encode_integer64_single", [](const CKKSEncoder& s, int64_t v, std::optional<ParmsID> id, Plaintext& d, PoolArg p) { s.encode_integer64_single(v, id, d, P(p)); },
             py::arg("value"), py::arg("parms_id"), py::arg("destination"), POOL)
        .def("encode_integer64_single_new", [](const CKKSEncoder& s, int64_t v, std::optional<ParmsID> id, PoolArg p) { return s.encode_integer64_single_new(v, id, P(p)); },
             py::arg("value"), py::arg("parms_id"), POOL)
        .def("encode_integer64_polynomial", [](const CKKSEncoder& s, const std::vector<int64_t>& v, std::optional<ParmsID> id, Plaintext& d, PoolArg p) { s.encode_integer64_polynomial(v, id, d, P(p)); },
             py::arg("values"), py::arg("parms_id"), py::arg("destination"), POOL)
        .def("encode_integer64_polynomial_new", [](const CKKSEncoder& s, const std::vector<int64_t>& v, std::optional<ParmsID> id, PoolArg p) { return s.encode_integer64_polynomial_new(v, id, P(p)); },
             py::arg("values"), py::arg("parms_id"), POOL);

    // PolynomialEncoderRing2k32 / 64 (pybind/src/polynomial_encoder_ring2k.cu); 128-bit elements have no numpy dtype and stay C++-only
    auto register_ring2k = [&m](auto tag, const char* name) {
        using T = decltype(tag);
        using Encoder = linear::PolynomialEncoderRing2k<T>;
        using Arr = py::array_t<T, py::array::c_style | py::array::forcecast>;
        py::class_<Encoder>(m, name)
            .def(py::init<HeContextPointer, size_t>(), py::arg("context"), py::arg("t_bit_length"))
            .def("context", &Encoder::context).def("on_device", &Encoder::on_device).def("t_bit_length", &Encoder::t_bit_length).def("slot_count", &Encoder::slot_count)
            .def("to_device_inplace", [](Encoder& s, PoolArg p) { s.to_device_inplace(P(p)); }, POOL)
            .def("scale_up", [](const Encoder& s, const Arr& v, std::optional<ParmsID> id, Plaintext& d, PoolArg p) { s.scale_up(std::vector<T>(v.data(), v.data() + v.size()), id, d, P(p)); },
                 py::arg("values"), py::arg("parms_id"), py::arg("destination"), POOL)
            .def("scale_up_new", [](const Encoder& s, const Arr& v, std::optional<ParmsID> id, PoolArg p) { return s.scale_up_new(std::vector<T>(v.data(), v.data() + v.size()), id, P(p)); },
                 py::arg("values"), py::arg("parms_id"), POOL)
            .def("centralize", [](const Encoder& s, const Arr& v, std::optional<ParmsID> id, Plaintext& d, PoolArg p) { s.centralize(std::vector<T>(v.data(), v.data() + v.size()), id, d, P(p)); },
                 py::arg("values"), py::arg("parms_id"), py::arg("destination"), POOL)
            .def("centralize_new", [](const Encoder& s, const Arr& v, std::optional<ParmsID> id, PoolArg p) { return s.centralize_new(std::vector<T>(v.data(), v.data() + v.size()), id, P(p)); },
                 py::arg("values"), py::arg("parms_id"), POOL)
            .def("scale_down_new", [](const Encoder& s, const Plaintext& pl, PoolArg p) { const std::vector<T> v = s.scale_down_new(pl, P(p)); return py::array_t<T>(v.size(), v.data()); },
                 py::arg("values"), POOL);
    };
    register_ring2k(uint32_t{}, "PolynomialEncoderRing2k32");
    register_ring2k(uint64_t{}, "PolynomialEncoderRing2k64");



    py::class_<LWECiphertext>(m, "LWECiphertext")
        .def(py::init<>())
        .def("clone", [](const LWECiphertext& s, PoolArg p) { return s.clone(P(p)); }, POOL)
        .def("on_device", &LWECiphertext::on_device)
        .def("parms_id", [](const LWECiphertext& s) { return s.parms_id(); })
        .def("coeff_modulus_size", [](const LWECiphertext& s) { return s.coeff_modulus_size(); })
        .def("poly_modulus_degree", [](const LWECiphertext& s) { return s.poly_modulus_degree(); })
        .def("scale", [](const LWECiphertext& s) { return s.scale(); })
        .def("assemble_lwe", [](const LWECiphertext& s, PoolArg p) { return s.assemble_lwe(P(p)); }, POOL)
        .def("address", [](const LWECiphertext& s) { return reinterpret_cast<uintptr_t>(&s); })
        .def("pool", [](const LWECiphertext& s) { return s.c1_dyn().pool(); })
        .def("device_index", [](const LWECiphertext& s) { return s.c1_dyn().pool() ? s.c1_dyn().pool()->get_device() : size_t(0); });

    py::class_<Evaluator> ev(m, "Evaluator");
    ev.def(py::init<HeContextPointer>()).def("context", &Evaluator::context).def("on_device", &Evaluator::on_device);
    EV_UNARY(ev, negate);
    ev.def("negate_inplace", [](const Evaluator& s, Ciphertext& a) { s.negate_inplace(a); }, py::arg("encrypted"));
    EV_BINARY(ev, add);
    EV_BINARY(ev, sub);
    EV_BINARY(ev, multiply);
    EV_UNARY(ev, square);
    EV_UNARY_INPLACE_POOL(ev, square);
    EV_KEYED(ev, relinearize, RelinKeys, "relin_keys");
    EV_KEYED(ev, apply_keyswitching, KSwitchKeys, "keyswitch_keys");
    EV_UNARY(ev, mod_switch_to_next);
    EV_UNARY_INPLACE_POOL(ev, mod_switch_to_next);
    EV_UNARY(ev, rescale_to_next);
    EV_UNARY_INPLACE_POOL(ev, rescale_to_next);
    EV_UNARY(ev, transform_to_ntt);
    EV_UNARY(ev, transform_from_ntt);
    ev.def("transform_to_ntt_inplace", [](const Evaluator& s, Ciphertext& a) { s.transform_to_ntt_inplace(a); }, py::arg("encrypted"));
    ev.def("transform_from_ntt_inplace", [](const Evaluator& s, Ciphertext& a) { s.transform_from_ntt_inplace(a); }, py::arg("encrypted"));
    ev.def("mod_switch_to", [](const Evaluator& s, const Ciphertext& a, const ParmsID& id, Ciphertext& d, PoolArg p) { s.mod_switch_to(a, id, d, P(p)); },
           py::arg("encrypted"), py::arg("parms_id"), py::arg("destination"), POOL);
    ev.def("mod_switch_to_inplace", [](const Evaluator& s, Ciphertext& a, const ParmsID& id, PoolArg p) { s.mod_switch_to_inplace(a, id, P(p)); }, py::arg("encrypted"), py::arg("parms_id"), POOL);
    ev.def("mod_switch_to_new", [](const Evaluator& s, const Ciphertext& a, const ParmsID& id, PoolArg p) { return s.mod_switch_to_new(a, id, P(p)); }, py::arg("encrypted"), py::arg("parms_id"), POOL);
    // ciphertext x plaintext
    ev.def("multiply_plain", [](const Evaluator& s, const Ciphertext& a, const Plaintext& w, Ciphertext& d, PoolArg p) { s.multiply_plain(a, w, d, P(p)); },
           py::arg("encrypted"), py::arg("plain"), py::arg("destination"), POOL);
    ev.def("multiply_plain_inplace", [](const Evaluator& s, Ciphertext& a, const Plaintext& w, PoolArg p) { s.multiply_plain_inplace(a, w, P(p)); }, py::arg("encrypted"), py::arg("plain"), POOL);
    ev.def("multiply_plain_new", [](const Evaluator& s, const Ciphertext& a, const Plaintext& w, PoolArg p) { return s.multiply_plain_new(a, w, P(p)); }, py::arg("encrypted"), py::arg("plain"), POOL);
    ev.def("transform_plain_to_ntt_new", [](const Evaluator& s, const Plaintext& w, const ParmsID& id, PoolArg p) { return s.transform_plain_to_ntt_new(w, id, P(p)); },
           py::arg("plain"), py::arg("parms_id"), POOL);
    ev.def("multiply_plain_accumulate", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const std::vector<Plaintext*>& w, const std::vector<Ciphertext*>& d, bool set_zero, PoolArg p) {
        s.multiply_plain_accumulate(const_ptrs(a), const_ptrs(w), d, set_zero, P(p));
    }, py::arg("encrypted"), py::arg("plain"), py::arg("destination"), py::arg("set_zero"), POOL);
    // batched forms (lists of objects)
    ev.def("add_batched", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const std::vector<Ciphertext*>& b, const std::vector<Ciphertext*>& d, PoolArg p) {
        s.add_batched(const_ptrs(a), const_ptrs(b), d, P(p)); }, py::arg("encrypted1"), py::arg("encrypted2"), py::arg("destination"), POOL);
    ev.def("multiply_batched", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const std::vector<Ciphertext*>& b, const std::vector<Ciphertext*>& d, PoolArg p) {
        s.multiply_batched(const_ptrs(a), const_ptrs(b), d, P(p)); }, py::arg("encrypted1"), py::arg("encrypted2"), py::arg("destination"), POOL);
    ev.def("relinearize_batched", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const RelinKeys& k, const std::vector<Ciphertext*>& d, PoolArg p) {
        s.relinearize_batched(const_ptrs(a), k, d, P(p)); }, py::arg("encrypted"), py::arg("relin_keys"), py::arg("destination"), POOL);
    ev.def("rescale_to_next_batched", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const std::vector<Ciphertext*>& d, PoolArg p) {
        s.rescale_to_next_batched(const_ptrs(a), d, P(p)); }, py::arg("encrypted"), py::arg("destination"), POOL);
    // additions to the reference's surface: multiply -> relinearize -> rescale_to_next behind one call (troy.h)
    ev.def("multiply_relinearize_rescale", [](const Evaluator& s, const Ciphertext& a, const Ciphertext& b, const RelinKeys& k, Ciphertext& d, PoolArg p) {
        s.multiply_relinearize_rescale(a, b, k, d, P(p)); }, py::arg("encrypted1"), py::arg("encrypted2"), py::arg("relin_keys"), py::arg("destination"), POOL);
    ev.def("multiply_relinearize_rescale_inplace", [](const Evaluator& s, Ciphertext& a, const Ciphertext& b, const RelinKeys& k, PoolArg p) {
        s.multiply_relinearize_rescale_inplace(a, b, k, P(p)); }, py::arg("encrypted1"), py::arg("encrypted2"), py::arg("relin_keys"), POOL);
    ev.def("multiply_relinearize_rescale_new", [](const Evaluator& s, const Ciphertext& a, const Ciphertext& b, const RelinKeys& k, PoolArg p) {
        return s.multiply_relinearize_rescale_new(a, b, k, P(p)); }, py::arg("encrypted1"), py::arg("encrypted2"), py::arg("relin_keys"), POOL);
    ev.def("multiply_relinearize_rescale_batched", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const std::vector<Ciphertext*>& b, const RelinKeys& k,
                                                      const std::vector<Ciphertext*>& d, PoolArg p) {
        s.multiply_relinearize_rescale_batched(const_ptrs(a), const_ptrs(b), k, d, P(p)); },
           py::arg("encrypted1"), py::arg("encrypted2"), py::arg("relin_keys"), py::arg("destination"), POOL);
    // Galois
    ev.def("apply_galois", [](const Evaluator& s, const Ciphertext& a, size_t g, const GaloisKeys& k, Ciphertext& d, PoolArg p) { s.apply_galois(a, g, k, d, P(p)); },
           py::arg("encrypted"), py::arg("galois_element"), py::arg("galois_keys"), py::arg("destination"), POOL);
    ev.def("apply_galois_new", [](const Evaluator& s, const Ciphertext& a, size_t g, const GaloisKeys& k, PoolArg p) { return s.apply_galois_new(a, g, k, P(p)); },
           py::arg("encrypted"), py::arg("galois_element"), py::arg("galois_keys"), POOL);
    ev.def("rotate_rows", [](const Evaluator& s, const Ciphertext& a, int st, const GaloisKeys& k, Ciphertext& d, PoolArg p) { s.rotate_rows(a, st, k, d, P(p)); },
           py::arg("encrypted"), py::arg("steps"), py::arg("galois_keys"), py::arg("destination"), POOL);
    ev.def("rotate_rows_inplace", [](const Evaluator& s, Ciphertext& a, int st, const GaloisKeys& k, PoolArg p) { s.rotate_rows_inplace(a, st, k, P(p)); },
           py::arg("encrypted"), py::arg("steps"), py::arg("galois_keys"), POOL);
    ev.def("rotate_rows_new", [](const Evaluator& s, const Ciphertext& a, int st, const GaloisKeys& k, PoolArg p) { return s.rotate_rows_new(a, st, k, P(p)); },
           py::arg("encrypted"), py::arg("steps"), py::arg("galois_keys"), POOL);
    EV_KEYED(ev, rotate_columns, GaloisKeys, "galois_keys");
    ev.def("rotate_vector_new", [](const Evaluator& s, const Ciphertext& a, int st, const GaloisKeys& k, PoolArg p) { return s.rotate_vector_new(a, st, k, P(p)); },
           py::arg("encrypted"), py::arg("steps"), py::arg("galois_keys"), POOL);
    ev.def("rotate_vector_inplace", [](const Evaluator& s, Ciphertext& a, int st, const GaloisKeys& k, PoolArg p) { s.rotate_vector_inplace(a, st, k, P(p)); },
           py::arg("encrypted"), py::arg("steps"), py::arg("galois_keys"), POOL);
    ev.def("complex_conjugate_new", [](const Evaluator& s, const Ciphertext& a, const GaloisKeys& k, PoolArg p) { return s.complex_conjugate_new(a, k, P(p)); },
           py::arg("encrypted"), py::arg("galois_keys"), POOL);
    // ciphertext +/- plaintext
#define EV_PLAIN(name)                                                                                                           \
    ev.def(#name, [](const Evaluator& s, const Ciphertext& a, const Plaintext& w, Ciphertext& d, PoolArg p) { s.name(a, w, d, P(p)); },  \
           py::arg("encrypted"), py::arg("plain"), py::arg("destination"), POOL);                                               \
    ev.def(#name "_inplace", [](const Evaluator& s, Ciphertext& a, const Plaintext& w, PoolArg p) { s.name##_inplace(a, w, P(p)); }, \
           py::arg("encrypted"), py::arg("plain"), POOL);                                                                       \
    ev.def(#name "_new", [](const Evaluator& s, const Ciphertext& a, const Plaintext& w, PoolArg p) { return s.name##_new(a, w, P(p)); }, \
           py::arg("encrypted"), py::arg("plain"), POOL)
    EV_PLAIN(add_plain);
    EV_PLAIN(sub_plain);
#undef EV_PLAIN
    ev.def("transform_plain_to_ntt", [](const Evaluator& s, const Plaintext& w, const ParmsID& id, Plaintext& d, PoolArg p) { s.transform_plain_to_ntt(w, id, d, P(p)); },
           py::arg("plain"), py::arg("parms_id"), py::arg("destination"), POOL);
    ev.def("transform_plain_to_ntt_inplace", [](const Evaluator& s, Plaintext& w, const ParmsID& id, PoolArg p) { s.transform_plain_to_ntt_inplace(w, id, P(p)); },
           py::arg("plain"), py::arg("parms_id"), POOL);
    ev.def("multiply_plain_new_batched", [](const Evaluator& s, const std::vector<Ciphertext*>& a, const std::vector<Plaintext*>& w, PoolArg p) {
        std::vector<Ciphertext> out(a.size());
        std::vector<Ciphertext*> op;
        for (Ciphertext& c : out) op.push_back(&c);
        s.multiply_plain_batched(const_ptrs(a), const_ptrs(w), op, P(p));
        return out;
    }, py::arg("encrypted"), py::arg("plain"), POOL);
    // rescale / plaintext modulus switching
    ev.def("rescale_to", [](const Evaluator& s, const Ciphertext& a, const ParmsID& id, Ciphertext& d, PoolArg p) { s.rescale_to(a, id, d, P(p)); },
           py::arg("encrypted"), py::arg("parms_id"), py::arg("destination"), POOL);
    ev.def("rescale_to_inplace", [](const Evaluator& s, Ciphertext& a, const ParmsID& id, PoolArg p) { s.rescale_to_inplace(a, id, P(p)); }, py::arg("encrypted"), py::arg("parms_id"), POOL);
    ev.def("rescale_to_new", [](const Evaluator& s, const Ciphertext& a, const ParmsID& id, PoolArg p) { return s.rescale_to_new(a, id, P(p)); }, py::arg("encrypted"), py::arg("parms_id"), POOL);
    ev.def("mod_switch_plain_to", [](const Evaluator& s, const Plaintext& a, const ParmsID& id, Plaintext& d, PoolArg p) { s.mod_switch_plain_to(a, id, d, P(p)); },
           py::arg("plain"), py::arg("parms_id"), py::arg("destination"), POOL);
    ev.def("mod_switch_plain_to_inplace", [](const Evaluator& s, Plaintext& a, const ParmsID& id, PoolArg p) { s.mod_switch_plain_to_inplace(a, id, P(p)); }, py::arg("plain"), py::arg("parms_id"), POOL);
    ev.def("mod_switch_plain_to_new", [](const Evaluator& s, const Plaintext& a, const ParmsID& id, PoolArg p) { return s.mod_switch_plain_to_new(a, id, P(p)); }, py::arg("plain"), py::arg("parms_id"), POOL);
    ev.def("mod_switch_plain_to_next", [](const Evaluator& s, const Plaintext& a, Plaintext& d, PoolArg p) { s.mod_switch_plain_to_next(a, d, P(p)); }, py::arg("plain"), py::arg("destination"), POOL);
    ev.def("mod_switch_plain_to_next_inplace", [](const Evaluator& s, Plaintext& a, PoolArg p) { s.mod_switch_plain_to_next_inplace(a, P(p)); }, py::arg("plain"), POOL);
    ev.def("mod_switch_plain_to_next_new", [](const Evaluator& s, const Plaintext& a, PoolArg p) { return s.mod_switch_plain_to_next_new(a, P(p)); }, py::arg("plain"), POOL);
    // Galois, remaining spellings
    ev.def("apply_galois_inplace", [](const Evaluator& s, Ciphertext& a, size_t g, const GaloisKeys& k, PoolArg p) { s.apply_galois_inplace(a, g, k, P(p)); },
           py::arg("encrypted"), py::arg("galois_element"), py::arg("galois_keys"), POOL);
    ev.def("rotate_vector", [](const Evaluator& s, const Ciphertext& a, int st, const GaloisKeys& k, Ciphertext& d, PoolArg p) { s.rotate_vector(a, st, k, d, P(p)); },
           py::arg("encrypted"), py::arg("steps"), py::arg("galois_keys"), py::arg("destination"), POOL);
    ev.def("complex_conjugate", [](const Evaluator& s, const Ciphertext& a, const GaloisKeys& k, Ciphertext& d, PoolArg p) { s.complex_conjugate(a, k, d, P(p)); },
           py::arg("encrypted"), py::arg("galois_keys"), py::arg("destination"), POOL);
    ev.def("complex_conjugate_inplace", [](const Evaluator& s, Ciphertext& a, const GaloisKeys& k, PoolArg p) { Ciphertext d; s.complex_conjugate(a, k, d, P(p)); a = std::move(d); },
           py::arg("encrypted"), py::arg("galois_keys"), POOL);
    ev.def("apply_galois_plain", [](const Evaluator& s, const Plaintext& a, size_t g, Plaintext& d, PoolArg p) { s.apply_galois_plain(a, g, d, P(p)); },
           py::arg("plain"), py::arg("galois_element"), py::arg("destination"), POOL);
    ev.def("apply_galois_plain_inplace", [](const Evaluator& s, Plaintext& a, size_t g, PoolArg p) { s.apply_galois_plain_inplace(a, g, P(p)); }, py::arg("plain"), py::arg("galois_element"), POOL);
    ev.def("apply_galois_plain_new", [](const Evaluator& s, const Plaintext& a, size_t g, PoolArg p) { return s.apply_galois_plain_new(a, g, P(p)); }, py::arg("plain"), py::arg("galois_element"), POOL);
    // LWE extraction and RLWE packing
    ev.def("extract_lwe_new", [](const Evaluator& s, const Ciphertext& a, size_t term, PoolArg p) { return s.extract_lwe_new(a, term, P(p)); }, py::arg("encrypted"), py::arg("term"), POOL);
    ev.def("assemble_lwe_new", [](const Evaluator& s, const LWECiphertext& l, PoolArg p) { return s.assemble_lwe_new(l, P(p)); }, py::arg("lwe_ciphertext"), POOL);
    ev.def("field_trace_inplace", [](const Evaluator& s, Ciphertext& a, const GaloisKeys& k, size_t logn, PoolArg p) { s.field_trace_inplace(a, k, logn, P(p)); },
           py::arg("encrypted"), py::arg("galois_keys"), py::arg("logn"), POOL);
    ev.def("divide_by_poly_modulus_degree_inplace", [](const Evaluator& s, Ciphertext& a, uint64_t mul) { s.divide_by_poly_modulus_degree_inplace(a, mul); },
           py::arg("encrypted"), py::arg("mul") = 1);
    ev.def("negacyclic_shift", [](const Evaluator& s, const Ciphertext& a, size_t shift, Ciphertext& d, PoolArg p) { s.negacyclic_shift(a, shift, d, P(p)); },
           py::arg("encrypted"), py::arg("shift"), py::arg("destination"), POOL);
    ev.def("negacyclic_shift_inplace", [](const Evaluator& s, Ciphertext& a, size_t shift, PoolArg p) { s.negacyclic_shift_inplace(a, shift, P(p)); }, py::arg("encrypted"), py::arg("shift"), POOL);
    ev.def("negacyclic_shift_new", [](const Evaluator& s, const Ciphertext& a, size_t shift, PoolArg p) { return s.negacyclic_shift_new(a, shift, P(p)); }, py::arg("encrypted"), py::arg("shift"), POOL);
    ev.def("pack_lwe_ciphertexts_new", [](const Evaluator& s, const std::vector<LWECiphertext*>& l, const GaloisKeys& k, PoolArg p, bool trace) {
        return s.pack_lwe_ciphertexts_new(const_ptrs(l), k, P(p), trace); }, py::arg("lwe_ciphertexts"), py::arg("automorphism_keys"), POOL, py::arg("apply_field_trace") = true);
    ev.def("pack_lwe_ciphertexts_new_batched", [](const Evaluator& s, const std::vector<std::vector<LWECiphertext*>>& groups, const GaloisKeys& k, PoolArg p, bool trace) {
        std::vector<std::vector<const LWECiphertext*>> g;
        for (const auto& grp : groups) g.push_back(const_ptrs(grp));
        return s.pack_lwe_ciphertexts_new_batched(g, k, P(p), trace); }, py::arg("lwe_groups"), py::arg("automorphism_keys"), POOL, py::arg("apply_field_trace") = true);
    ev.def("pack_rlwe_ciphertexts_new", [](const Evaluator& s, const std::vector<Ciphertext*>& c, const GaloisKeys& k, size_t shift, size_t in_iv, size_t out_iv, PoolArg p, bool trace) {
        return s.pack_rlwe_ciphertexts_new(const_ptrs(c), k, shift, in_iv, out_iv, P(p), trace); },
        py::arg("rlwe_ciphertexts"), py::arg("automorphism_keys"), py::arg("shift"), py::arg("input_interval"), py::arg("output_interval"), POOL, py::arg("apply_field_trace") = true);
    ev.def("pack_rlwe_ciphertexts_new_batched", [](const Evaluator& s, const std::vector<std::vector<Ciphertext*>>& groups, const GaloisKeys& k, size_t shift, size_t in_iv, size_t out_iv,
                                                   PoolArg p, bool trace) {
        std::vector<std::vector<const Ciphertext*>> g;
        for (const auto& grp : groups) g.push_back(const_ptrs(grp));
        return s.pack_rlwe_ciphertexts_new_batched(g, k, shift, in_iv, out_iv, P(p), trace); },
        py::arg("rlwe_groups"), py::arg("automorphism_keys"), py::arg("shift"), py::arg("input_interval"), py::arg("output_interval"), POOL, py::arg("apply_field_trace") = true);

    // ---- linear-algebra applications (pybind/src/matmul_helper.cu, conv2d_helper.cu): numpy arrays in, numpy arrays out ----
    using linear::Cipher2d; using linear::Conv2dHelper; using linear::MatmulHelper; using linear::MatmulObjective; using linear::Plain2d;
    auto vec_u64 = [](const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& a) { return std::vector<uint64_t>(a.data(), a.data() + a.size()); };
    auto vec_f64 = [](const py::array_t<double, py::array::c_style | py::array::forcecast>& a) { return std::vector<double>(a.data(), a.data() + a.size()); };
    auto arr_u64 = [](const std::vector<uint64_t>& v) { return py::array_t<uint64_t>(v.size(), v.data()); };
    auto arr_f64 = [](const std::vector<double>& v) { return py::array_t<double>(v.size(), v.data()); };
    auto need = [](size_t got, size_t want, const char* what) { if (got != want) throw std::invalid_argument(std::string("[") + what + "] array size does not match the helper's dimensions."); };

    py::class_<Plain2d>(m, "Plain2d")
        .def(py::init<>()).def("size", &Plain2d::size).def("rows", &Plain2d::rows).def("columns", &Plain2d::columns)
        .def("clone", [](const Plain2d& s, PoolArg p) { return s.clone(P(p)); }, POOL)
        .def("encrypt_symmetric", [](const Plain2d& s, const Encryptor& e, PoolArg p) { return s.encrypt_symmetric(e, P(p)); }, py::arg("encryptor"), POOL)
        .def("encrypt_asymmetric", [](const Plain2d& s, const Encryptor& e, PoolArg p) { return s.encrypt_asymmetric(e, P(p)); }, py::arg("encryptor"), POOL)
        .def("get", [](Plain2d& s, size_t i, size_t j) -> Plaintext& { return s[i].at(j); }, py::arg("i"), py::arg("j"), py::return_value_policy::reference_internal);
    py::class_<Cipher2d>(m, "Cipher2d")
        .def(py::init<>()).def("size", &Cipher2d::size).def("rows", &Cipher2d::rows).def("columns", &Cipher2d::columns)
        .def("clone", [](const Cipher2d& s, PoolArg p) { return s.clone(P(p)); }, POOL)
        .def("expand_seed", &Cipher2d::expand_seed)
        .def("save", [](const Cipher2d& s, HeContextPointer c, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.save(os, c, mode); }); }, py::arg("context"), MODE)
        .def("load", [](Cipher2d& s, const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; s.load(is, c, P(p)); }, py::arg("str"), py::arg("context"), POOL)
        .def_static("load_new", [](const py::bytes& b, HeContextPointer c, PoolArg p) { std::istringstream is{std::string(b)}; return Cipher2d::load_new(is, c, P(p)); },
                    py::arg("str"), py::arg("context"), POOL)
        .def("serialized_size_upperbound", [](const Cipher2d& s, HeContextPointer c, CompressionMode mode) { return s.serialized_size_upperbound(c, mode); }, py::arg("context"), MODE)
        .def("mod_switch_to_next_inplace", [](Cipher2d& s, const Evaluator& e, PoolArg p) { s.mod_switch_to_next_inplace(e, P(p)); }, py::arg("evaluator"), POOL)
        .def("mod_switch_to_next", [](const Cipher2d& s, const Evaluator& e, PoolArg p) { return s.mod_switch_to_next(e, P(p)); }, py::arg("evaluator"), POOL)
        .def("relinearize_inplace", [](Cipher2d& s, const Evaluator& e, const RelinKeys& k, PoolArg p) { s.relinearize_inplace(e, k, P(p)); }, py::arg("evaluator"), py::arg("relin_keys"), POOL)
        .def("relinearize", [](const Cipher2d& s, const Evaluator& e, const RelinKeys& k, PoolArg p) { return s.relinearize(e, k, P(p)); }, py::arg("evaluator"), py::arg("relin_keys"), POOL)
        .def("add", [](const Cipher2d& s, const Evaluator& e, const Cipher2d& o, PoolArg p) { return s.add(e, o, P(p)); }, py::arg("evaluator"), py::arg("other"), POOL)
        .def("add_inplace", [](Cipher2d& s, const Evaluator& e, const Cipher2d& o, PoolArg p) { s.add_inplace(e, o, P(p)); }, py::arg("evaluator"), py::arg("other"), POOL)
        .def("add_plain", [](const Cipher2d& s, const Evaluator& e, const Plain2d& o, PoolArg p) { return s.add_plain(e, o, P(p)); }, py::arg("evaluator"), py::arg("plain"), POOL)
        .def("add_plain_inplace", [](Cipher2d& s, const Evaluator& e, const Plain2d& o, PoolArg p) { s.add_plain_inplace(e, o, P(p)); }, py::arg("evaluator"), py::arg("plain"), POOL)
        .def("sub", [](const Cipher2d& s, const Evaluator& e, const Cipher2d& o, PoolArg p) { return s.sub(e, o, P(p)); }, py::arg("evaluator"), py::arg("other"), POOL)
        .def("sub_inplace", [](Cipher2d& s, const Evaluator& e, const Cipher2d& o, PoolArg p) { s.sub_inplace(e, o, P(p)); }, py::arg("evaluator"), py::arg("other"), POOL)
        .def("sub_plain", [](const Cipher2d& s, const Evaluator& e, const Plain2d& o, PoolArg p) { return s.sub_plain(e, o, P(p)); }, py::arg("evaluator"), py::arg("plain"), POOL)
        .def("sub_plain_inplace", [](Cipher2d& s, const Evaluator& e, const Plain2d& o, PoolArg p) { s.sub_plain_inplace(e, o, P(p)); }, py::arg("evaluator"), py::arg("plain"), POOL)
        .def("decrypt", [](const Cipher2d& s, const Decryptor& d, PoolArg p) { return s.decrypt(d, P(p)); }, py::arg("decryptor"), POOL)
        .def("get", [](Cipher2d& s, size_t i, size_t j) -> Ciphertext& { return s[i].at(j); }, py::arg("i"), py::arg("j"), py::return_value_policy::reference_internal);
    py::enum_<MatmulObjective>(m, "MatmulObjective").value("EncryptLeft", MatmulObjective::EncryptLeft).value("EncryptRight", MatmulObjective::EncryptRight)
        .value("Crossed", MatmulObjective::Crossed);

    py::class_<MatmulHelper> mh(m, "MatmulHelper");
    mh.def(py::init([](size_t b, size_t i, size_t o, size_t n, MatmulObjective obj, bool pack, PoolArg p) { return MatmulHelper(b, i, o, n, obj, pack, P(p)); }),
           py::arg("batch_size"), py::arg("input_dims"), py::arg("output_dims"), py::arg("slot_count"), py::arg("objective") = MatmulObjective::EncryptLeft,
           py::arg("pack_lwe") = true, POOL)
        .def("set_pool", &MatmulHelper::set_pool)
        .def("batch_size", [](const MatmulHelper& s) { return s.batch_size; }).def("input_dims", [](const MatmulHelper& s) { return s.input_dims; })
        .def("output_dims", [](const MatmulHelper& s) { return s.output_dims; }).def("slot_count", [](const MatmulHelper& s) { return s.slot_count; })
        .def("objective", [](const MatmulHelper& s) { return s.objective; }).def("pack_lwe", [](const MatmulHelper& s) { return s.pack_lwe; })
        .def("batch_block", [](const MatmulHelper& s) { return s.batch_block; }).def("input_block", [](const MatmulHelper& s) { return s.input_block; })
        .def("output_block", [](const MatmulHelper& s) { return s.output_block; })
        .def("matmul", &MatmulHelper::matmul).def("matmul_reverse", &MatmulHelper::matmul_reverse).def("matmul_cipher", &MatmulHelper::matmul_cipher)
        .def("pack_outputs", &MatmulHelper::pack_outputs)
        .def("serialize_outputs", [](const MatmulHelper& s, const Evaluator& e, const Cipher2d& x, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.serialize_outputs(e, x, os, mode); }); },
             py::arg("evaluator"), py::arg("x"), MODE)
        .def("deserialize_outputs", [](const MatmulHelper& s, const Evaluator& e, const py::bytes& b) { std::istringstream is{std::string(b)}; return s.deserialize_outputs(e, is); }, py::arg("evaluator"), py::arg("str"));
    auto mm_enc_w = [=](const MatmulHelper& s, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& w) {
        need(w.size(), s.input_dims * s.output_dims, "MatmulHelper::encode_weights"); return s.encode_weights_uint64s(enc, vec_u64(w).data()); };
    auto mm_enc_x = [=](const MatmulHelper& s, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& x) {
        need(x.size(), s.batch_size * s.input_dims, "MatmulHelper::encode_inputs"); return s.encode_inputs_uint64s(enc, vec_u64(x).data()); };
    auto mm_cry_w = [=](const MatmulHelper& s, const Encryptor& e, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& w) {
        need(w.size(), s.input_dims * s.output_dims, "MatmulHelper::encrypt_weights"); return s.encrypt_weights_uint64s(e, enc, vec_u64(w).data()); };
    auto mm_cry_x = [=](const MatmulHelper& s, const Encryptor& e, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& x) {
        need(x.size(), s.batch_size * s.input_dims, "MatmulHelper::encrypt_inputs"); return s.encrypt_inputs_uint64s(e, enc, vec_u64(x).data()); };
    auto mm_enc_y = [=](const MatmulHelper& s, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& y) {
        need(y.size(), s.batch_size * s.output_dims, "MatmulHelper::encode_outputs"); return s.encode_outputs_uint64s(enc, vec_u64(y).data()); };
    auto mm_dec_y = [=](const MatmulHelper& s, const BatchEncoder& enc, const Decryptor& d, const Cipher2d& y) { return arr_u64(s.decrypt_outputs_uint64s(enc, d, y)); };
    for (const char* suffix : {"", "_uint64s"}) {
        mh.def((std::string("encode_weights") + suffix).c_str(), mm_enc_w).def((std::string("encode_inputs") + suffix).c_str(), mm_enc_x)
          .def((std::string("encrypt_weights") + suffix).c_str(), mm_cry_w).def((std::string("encrypt_inputs") + suffix).c_str(), mm_cry_x)
          .def((std::string("encode_outputs") + suffix).c_str(), mm_enc_y).def((std::string("decrypt_outputs") + suffix).c_str(), mm_dec_y);
    }
    mh.def("encode_weights_doubles", [=](const MatmulHelper& s, const CKKSEncoder& enc, const py::array_t<double, py::array::c_style | py::array::forcecast>& w, std::optional<ParmsID> id, double scale) {
            need(w.size(), s.input_dims * s.output_dims, "MatmulHelper::encode_weights"); return s.encode_weights_doubles(enc, vec_f64(w).data(), id, scale); })
      .def("encode_inputs_doubles", [=](const MatmulHelper& s, const CKKSEncoder& enc, const py::array_t<double, py::array::c_style | py::array::forcecast>& x, std::optional<ParmsID> id, double scale) {
            need(x.size(), s.batch_size * s.input_dims, "MatmulHelper::encode_inputs"); return s.encode_inputs_doubles(enc, vec_f64(x).data(), id, scale); })
      .def("encrypt_inputs_doubles", [=](const MatmulHelper& s, const Encryptor& e, const CKKSEncoder& enc, const py::array_t<double, py::array::c_style | py::array::forcecast>& x, std::optional<ParmsID> id, double scale) {
            need(x.size(), s.batch_size * s.input_dims, "MatmulHelper::encrypt_inputs"); return s.encrypt_inputs_doubles(e, enc, vec_f64(x).data(), id, scale); })
      .def("encrypt_weights_doubles", [=](const MatmulHelper& s, const Encryptor& e, const CKKSEncoder& enc, const py::array_t<double, py::array::c_style | py::array::forcecast>& w, std::optional<ParmsID> id, double scale) {
            need(w.size(), s.input_dims * s.output_dims, "MatmulHelper::encrypt_weights"); return s.encrypt_weights_doubles(e, enc, vec_f64(w).data(), id, scale); })
      .def("encode_outputs_doubles", [=](const MatmulHelper& s, const CKKSEncoder& enc, const py::array_t<double, py::array::c_style | py::array::forcecast>& y, std::optional<ParmsID> id, double scale) {
            need(y.size(), s.batch_size * s.output_dims, "MatmulHelper::encode_outputs"); return s.encode_outputs_doubles(enc, vec_f64(y).data(), id, scale); })
      .def("decrypt_outputs_doubles", [=](const MatmulHelper& s, const CKKSEncoder& enc, const Decryptor& d, const Cipher2d& y) { return arr_f64(s.decrypt_outputs_doubles(enc, d, y)); });

    py::class_<Conv2dHelper> ch(m, "Conv2dHelper");
    ch.def(py::init([](size_t b, size_t ic, size_t oc, size_t h, size_t w, size_t kh, size_t kw, size_t n, MatmulObjective obj, PoolArg p) { return Conv2dHelper(b, ic, oc, h, w, kh, kw, n, obj, P(p)); }),
           py::arg("batch_size"), py::arg("input_channels"), py::arg("output_channels"), py::arg("image_height"), py::arg("image_width"), py::arg("kernel_height"),
           py::arg("kernel_width"), py::arg("poly_degree"), py::arg("objective") = MatmulObjective::EncryptLeft, POOL)
        .def("batch_size", [](const Conv2dHelper& s) { return s.batch_size; }).def("input_channels", [](const Conv2dHelper& s) { return s.input_channels; })
        .def("output_channels", [](const Conv2dHelper& s) { return s.output_channels; }).def("image_height", [](const Conv2dHelper& s) { return s.image_height; })
        .def("image_width", [](const Conv2dHelper& s) { return s.image_width; }).def("kernel_height", [](const Conv2dHelper& s) { return s.kernel_height; })
        .def("kernel_width", [](const Conv2dHelper& s) { return s.kernel_width; }).def("slot_count", [](const Conv2dHelper& s) { return s.slot_count; })
        .def("objective", [](const Conv2dHelper& s) { return s.objective; }).def("batch_block", [](const Conv2dHelper& s) { return s.batch_block; })
        .def("input_channel_block", [](const Conv2dHelper& s) { return s.input_channel_block; }).def("output_channel_block", [](const Conv2dHelper& s) { return s.output_channel_block; })
        .def("image_height_block", [](const Conv2dHelper& s) { return s.image_height_block; }).def("image_width_block", [](const Conv2dHelper& s) { return s.image_width_block; })
        .def("conv2d", &Conv2dHelper::conv2d).def("conv2d_reverse", &Conv2dHelper::conv2d_reverse).def("conv2d_cipher", &Conv2dHelper::conv2d_cipher)
        .def("serialize_outputs", [](const Conv2dHelper& s, const Evaluator& e, const Cipher2d& x, CompressionMode mode) { return to_bytes([&](std::ostream& os) { s.serialize_outputs(e, x, os, mode); }); },
             py::arg("evaluator"), py::arg("x"), MODE)
        .def("deserialize_outputs", [](const Conv2dHelper& s, const Evaluator& e, const py::bytes& b) { std::istringstream is{std::string(b)}; return s.deserialize_outputs(e, is); }, py::arg("evaluator"), py::arg("str"));
    auto cv_out = [](const Conv2dHelper& s) { return s.batch_size * s.output_channels * (s.image_height - s.kernel_height + 1) * (s.image_width - s.kernel_width + 1); };
    auto cv_enc_w = [=](const Conv2dHelper& s, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& w) {
        need(w.size(), s.output_channels * s.input_channels * s.kernel_height * s.kernel_width, "Conv2dHelper::encode_weights"); return s.encode_weights_uint64s(enc, vec_u64(w).data()); };
    auto cv_enc_x = [=](const Conv2dHelper& s, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& x) {
        need(x.size(), s.batch_size * s.input_channels * s.image_height * s.image_width, "Conv2dHelper::encode_inputs"); return s.encode_inputs_uint64s(enc, vec_u64(x).data()); };
    auto cv_cry_w = [=](const Conv2dHelper& s, const Encryptor& e, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& w) {
        need(w.size(), s.output_channels * s.input_channels * s.kernel_height * s.kernel_width, "Conv2dHelper::encrypt_weights"); return s.encrypt_weights_uint64s(e, enc, vec_u64(w).data()); };
    auto cv_cry_x = [=](const Conv2dHelper& s, const Encryptor& e, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& x) {
        need(x.size(), s.batch_size * s.input_channels * s.image_height * s.image_width, "Conv2dHelper::encrypt_inputs"); return s.encrypt_inputs_uint64s(e, enc, vec_u64(x).data()); };
    auto cv_enc_y = [=](const Conv2dHelper& s, const BatchEncoder& enc, const py::array_t<uint64_t, py::array::c_style | py::array::forcecast>& y) {
        need(y.size(), cv_out(s), "Conv2dHelper::encode_outputs"); return s.encode_outputs_uint64s(enc, vec_u64(y).data()); };
    auto cv_dec_y = [=](const Conv2dHelper& s, const BatchEncoder& enc, const Decryptor& d, const Cipher2d& y) { return arr_u64(s.decrypt_outputs_uint64s(enc, d, y)); };
    for (const char* suffix : {"", "_uint64s"}) {
        ch.def((std::string("encode_weights") + suffix).c_str(), cv_enc_w).def((std::string("encode_inputs") + suffix).c_str(), cv_enc_x)
          .def((std::string("encrypt_weights") + suffix).c_str(), cv_cry_w).def((std::string("encrypt_inputs") + suffix).c_str(), cv_cry_x)
          .def((std::string("encode_outputs") + suffix).c_str(), cv_enc_y).def((std::string("decrypt_outputs") + suffix).c_str(), cv_dec_y);
    }
    using ArrF = py::array_t<double, py::array::c_style | py::array::forcecast>;
    auto cv_w = [](const Conv2dHelper& s) { return s.output_channels * s.input_channels * s.kernel_height * s.kernel_width; };
    auto cv_x = [](const Conv2dHelper& s) { return s.batch_size * s.input_channels * s.image_height * s.image_width; };
    ch.def("set_pool", &Conv2dHelper::set_pool)
      .def("encode_weights_doubles", [=](const Conv2dHelper& s, const CKKSEncoder& enc, const ArrF& w, std::optional<ParmsID> id, double scale) {
            need(w.size(), cv_w(s), "Conv2dHelper::encode_weights"); return s.encode_weights_doubles(enc, vec_f64(w).data(), id, scale); })
      .def("encode_inputs_doubles", [=](const Conv2dHelper& s, const CKKSEncoder& enc, const ArrF& x, std::optional<ParmsID> id, double scale) {
            need(x.size(), cv_x(s), "Conv2dHelper::encode_inputs"); return s.encode_inputs_doubles(enc, vec_f64(x).data(), id, scale); })
      .def("encrypt_weights_doubles", [=](const Conv2dHelper& s, const Encryptor& e, const CKKSEncoder& enc, const ArrF& w, std::optional<ParmsID> id, double scale) {
            need(w.size(), cv_w(s), "Conv2dHelper::encrypt_weights"); return s.encrypt_weights_doubles(e, enc, vec_f64(w).data(), id, scale); })
      .def("encrypt_inputs_doubles", [=](const Conv2dHelper& s, const Encryptor& e, const CKKSEncoder& enc, const ArrF& x, std::optional<ParmsID> id, double scale) {
            need(x.size(), cv_x(s), "Conv2dHelper::encrypt_inputs"); return s.encrypt_inputs_doubles(e, enc, vec_f64(x).data(), id, scale); })
      .def("encode_outputs_doubles", [=](const Conv2dHelper& s, const CKKSEncoder& enc, const ArrF& y, std::optional<ParmsID> id, double scale) {
            need(y.size(), cv_out(s), "Conv2dHelper::encode_outputs"); return s.encode_outputs_doubles(enc, vec_f64(y).data(), id, scale); })
      .def("decrypt_outputs_doubles", [=](const Conv2dHelper& s, const CKKSEncoder& enc, const Decryptor& d, const Cipher2d& y) { return arr_f64(s.decrypt_outputs_doubles(enc, d, y)); });
    // ring-2^k forms of the helpers, registered per element width as <name>_ring2k32 / _ring2k64 (pybind/src/matmul_helper.cu:7-80, conv2d_helper.cu:9-85)
    auto register_helper_ring2k = [&](auto tag, const std::string& bits) {
        using T = decltype(tag);
        using Encoder = linear::PolynomialEncoderRing2k<T>;
        using Arr = py::array_t<T, py::array::c_style | py::array::forcecast>;
        auto vec = [](const Arr& a) { return std::vector<T>(a.data(), a.data() + a.size()); };
        auto out = [](const std::vector<T>& v) { return py::array_t<T>(v.size(), v.data()); };
        mh.def(("encode_weights_ring2k" + bits).c_str(), [=](const MatmulHelper& s, const Encoder& enc, const Arr& w, std::optional<ParmsID> id) {
                need(w.size(), s.input_dims * s.output_dims, "MatmulHelper::encode_weights"); return s.encode_weights_ring2k<T>(enc, vec(w).data(), id); })
          .def(("encode_inputs_ring2k" + bits).c_str(), [=](const MatmulHelper& s, const Encoder& enc, const Arr& x, std::optional<ParmsID> id) {
                need(x.size(), s.batch_size * s.input_dims, "MatmulHelper::encode_inputs"); return s.encode_inputs_ring2k<T>(enc, vec(x).data(), id); })
          .def(("encode_outputs_ring2k" + bits).c_str(), [=](const MatmulHelper& s, const Encoder& enc, const Arr& y, std::optional<ParmsID> id) {
                need(y.size(), s.batch_size * s.output_dims, "MatmulHelper::encode_outputs"); return s.encode_outputs_ring2k<T>(enc, vec(y).data(), id); })
          .def(("encrypt_weights_ring2k" + bits).c_str(), [=](const MatmulHelper& s, const Encryptor& e, const Encoder& enc, const Arr& w, std::optional<ParmsID> id) {
                need(w.size(), s.input_dims * s.output_dims, "MatmulHelper::encrypt_weights"); return s.encrypt_weights_ring2k<T>(e, enc, vec(w).data(), id); })
          .def(("encrypt_inputs_ring2k" + bits).c_str(), [=](const MatmulHelper& s, const Encryptor& e, const Encoder& enc, const Arr& x, std::optional<ParmsID> id) {
                need(x.size(), s.batch_size * s.input_dims, "MatmulHelper::encrypt_inputs"); return s.encrypt_inputs_ring2k<T>(e, enc, vec(x).data(), id); })
          .def(("decrypt_outputs_ring2k" + bits).c_str(), [=](const MatmulHelper& s, const Encoder& enc, const Decryptor& d, const Cipher2d& y) { return out(s.decrypt_outputs_ring2k<T>(enc, d, y)); });
        ch.def(("encode_weights_ring2k" + bits).c_str(), [=](const Conv2dHelper& s, const Encoder& enc, const Arr& w, std::optional<ParmsID> id) {
                need(w.size(), cv_w(s), "Conv2dHelper::encode_weights"); return s.encode_weights_ring2k<T>(enc, vec(w).data(), id); })
          .def(("encode_inputs_ring2k" + bits).c_str(), [=](const Conv2dHelper& s, const Encoder& enc, const Arr& x, std::optional<ParmsID> id) {
                need(x.size(), cv_x(s), "Conv2dHelper::encode_inputs"); return s.encode_inputs_ring2k<T>(enc, vec(x).data(), id); })
          .def(("encode_outputs_ring2k" + bits).c_str(), [=](const Conv2dHelper& s, const Encoder& enc, const Arr& y, std::optional<ParmsID> id) {
                need(y.size(), cv_out(s), "Conv2dHelper::encode_outputs"); return s.encode_outputs_ring2k<T>(enc, vec(y).data(), id); })
          .def(("encrypt_weights_ring2k" + bits).c_str(), [=](const Conv2dHelper& s, const Encryptor& e, const Encoder& enc, const Arr& w, std::optional<ParmsID> id) {
                need(w.size(), cv_w(s), "Conv2dHelper::encrypt_weights"); return s.encrypt_weights_ring2k<T>(e, enc, vec(w).data(), id); })
          .def(("encrypt_inputs_ring2k" + bits).c_str(), [=](const Conv2dHelper& s, const Encryptor& e, const Encoder& enc, const Arr& x, std::optional<ParmsID> id) {
                need(x.size(), cv_x(s), "Conv2dHelper::encrypt_inputs"); return s.encrypt_inputs_ring2k<T>(e, enc, vec(x).data(), id); })
          .def(("decrypt_outputs_ring2k" + bits).c_str(), [=](const Conv2dHelper& s, const Encoder& enc, const Decryptor& d, const Cipher2d& y) { return out(s.decrypt_outputs_ring2k<T>(enc, d, y)); });
    };
    register_helper_ring2k(uint32_t{}, "32");
    register_helper_ring2k(uint64_t{}, "64");
}
