"""MI355X-native Aho-Corasick matcher with the ahocorasick_rs API (placeholder
until the C++ extension is built; see capi.py for the C-ABI binding)."""
