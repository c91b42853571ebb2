"""Lists the method names the reference's Python module registers per class (pybind/src/*.cu: every `.def("name"` / `.def_static("name"`)
into tests/golden/pytroy_surface.json.  Runs in the build container only (/root/reference is not present on the GPU box); the JSON
is data -- names, no code -- and tests/test_pytroy.py::test_surface_is_complete checks the built module against it."""
import json
import os
import re

REF = "/root/reference/pybind/src"
CLASSES = {
    "batch_encoder": ["BatchEncoder"], "ciphertext": ["Ciphertext"], "ckks_encoder": ["CKKSEncoder"], "conv2d_helper": ["Conv2dHelper"],
    "decryptor": ["Decryptor"], "encryptor": ["Encryptor"], "evaluator": ["Evaluator"], "key_generator": ["KeyGenerator"],
    "lwe_ciphertext": ["LWECiphertext"], "plaintext": ["Plaintext"], "public_key": ["PublicKey"], "secret_key": ["SecretKey"],
    "random_generator": ["RandomGenerator"], "polynomial_encoder_ring2k": ["PolynomialEncoderRing2k32", "PolynomialEncoderRing2k64"],
    "kswitch_keys": ["KSwitchKeys", "RelinKeys", "GaloisKeys"], "matmul_helper": ["MatmulHelper", "Plain2d", "Cipher2d"],
    "modulus": ["Modulus", "CoeffModulus", "PlainModulus"], "he_context": ["HeContext", "ContextData"],
    "encryption_parameters": ["EncryptionParameters", "ParmsID"], "basics": ["MemoryPool"], "binder": [],
}

out = {}
for f, classes in sorted(CLASSES.items()):
    src = open(os.path.join(REF, f + ".cu")).read()
    names = set(re.findall(r'\.def(?:_static)?\(\s*"([A-Za-z0-9_]+)"', src))
    # names assembled at registration time: .def((std::string("encode_weights_ring2k") + bitwidth_name).c_str(), ...) for "32" and "64"
    for stem in re.findall(r'std::string\("([A-Za-z0-9_]+)"\)\s*\+\s*bitwidth_name', src):
        names |= {stem + "32", stem + "64"}
    names = sorted(names)
    # keyword-argument lists, one per overload that names its arguments (py::arg("...") and the header's argument macros)
    macros = {"MEMORY_POOL_ARGUMENT": "pool", "COMPRESSION_MODE_ARGUMENT": "mode", "OPTIONAL_PARMS_ID_ARGUMENT": "parms_id"}
    signatures = {}
    for mm in re.finditer(r'\.def(?:_static)?\(\s*"([A-Za-z0-9_]+)"(.*?)(?=\n\s*\.def|\n\s*;)', src, re.S):
        name, body = mm.group(1), mm.group(2)
        tail = body[body.rfind("}") + 1:] if "}" in body else body
        args = [t.group(1) or macros[t.group(2)]
                for t in re.finditer(r'py::arg\("([a-z_0-9]+)"\)|(MEMORY_POOL_ARGUMENT|COMPRESSION_MODE_ARGUMENT|OPTIONAL_PARMS_ID_ARGUMENT)', tail)]
        if args and args not in signatures.setdefault(name, []):
            signatures[name].append(args)
    out[f] = {"classes": classes, "names": names, "signatures": signatures}   # several classes per file: a name must exist on one of them
out["binder"]["classes"] = ["<module>"]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pytroy_surface.json"), "w"), indent=1, sort_keys=True)
print(sum(len(v["names"]) for v in out.values()), "names")
