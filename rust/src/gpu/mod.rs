//! GPU batch witness generation for `verify_secp256k1_message_circuit` (binding of libp2e_hip.so).
//! Source only: not compiled in the repository that ships it (no Rust toolchain there).
pub mod bind;
pub mod ffi;
