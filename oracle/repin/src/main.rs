// oracle/repin/src/main.rs -- prints, from the REAL conroi/lcpc crates, the values that this repository's golden
// fixtures (tests/golden/commit_cases.json, made by the Python restatement) claim for the same inputs: ALL nine cases
// (four fields, rates 1/2, 1/4 and 38/39 with new_from_dims, SdigCode3 and SdigCode5, Brakedown proofs), with the same
// keys the fixture holds.  `python oracle/repin/compare.py < output` reports the first differing line.
// Equal output  =>  every third-party convention marked [3P] in oracle/ is pinned at once:
//   ff_derive Montgomery form / to_repr / From<u64> / Field::random, fffft root + output order, BLAKE3 leaf format,
//   Merkle layout, merlin transcript, ChaCha20 (from_seed and seed_from_u64), rand Uniform (column draws and matgen),
//   sprs-independent expander encoding, bincode.
// Not compiled in this repo's image (no Rust toolchain); written against the API in /root/reference:
//   LcCommit::commit / prove / get_root + Serialize (lcpc-2d/src/lib.rs:186-312), LcEvalProof::verify (lib.rs:518-527),
//   LigeroEncodingRho::{new, new_from_dims} (ligero lib.rs:121-148), SdigEncodingS::new (brakedown lib.rs:103-110),
//   the test transcript set-up (ligero tests.rs:243-245).
use blake3::Hasher as Blake3;
use ff::{Field, PrimeField};
use lcpc_2d::{LcCommit, LcEncoding};
use lcpc_brakedown_pc::codespec::{SdigCode3, SdigCode5};
use lcpc_brakedown_pc::SdigEncodingS;
use lcpc_ligero_pc::LigeroEncodingRho;
use lcpc_test_fields::{ft127::Ft127, ft191::Ft191, ft255::Ft255, ft63::Ft63};
use merlin::Transcript;
use rand_chacha::{rand_core::SeedableRng, ChaCha20Rng};
use sha2::{Digest as _, Sha256};
use typenum::{U1, U2, U38, U39, U4};

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}
fn hex_be(le: &[u8]) -> String {
    // the fixture prints integers: big-endian hex without leading zeros
    let s: String = le.iter().rev().map(|x| format!("{:02x}", x)).collect();
    let t = s.trim_start_matches('0');
    format!("0x{}", if t.is_empty() { "0" } else { t })
}

fn powers<F: Field>(x: F, n: usize) -> Vec<F> {
    let mut v = Vec::with_capacity(n);
    let mut cur = F::one();
    for _ in 0..n {
        v.push(cur);
        cur *= x;
    }
    v
}

// inputs of tests/golden/make_golden.py: "iota" c_i = i + 1; "rand" ChaCha20Rng::from_seed([seed; 32]) + Field::random
fn coeffs<F: PrimeField>(n: usize, kind: &str, seed: u8) -> Vec<F> {
    if kind == "iota" {
        (1..=n as u64).map(F::from).collect()
    } else {
        let mut rng = ChaCha20Rng::from_seed([seed; 32]);
        (0..n).map(|_| F::random(&mut rng)).collect()
    }
}

fn case<F, E>(name: &str, coeffs: &[F], enc: &E, with_proof: bool)
where
    F: PrimeField + serde::Serialize + serde::de::DeserializeOwned,
    E: LcEncoding<F = F>,
{
    let comm = LcCommit::<Blake3, E>::commit(coeffs, enc).unwrap();
    let root = comm.get_root();
    let (nr, np, nc) = (comm.get_n_rows(), comm.get_n_per_row(), comm.get_n_cols());
    println!("{} dims {} {} {}", name, nr, np, nc);
    println!("{} n_col_opens {} n_degree_tests {}", name, enc.get_n_col_opens(), enc.get_n_degree_tests());
    println!("{} root {}", name, hex(root.as_ref()));
    // LcCommit's fields are private: read comm and hashes out of its serde form (lib.rs:186-197):
    // u64 len, comm elements (raw Montgomery limbs) | u64 len, coeffs | n_rows, n_cols, n_per_row | u64 len, (u64 32, digest)*
    let dump = bincode::serialize(&comm).unwrap();
    // ... and the serde form itself (the bytes lcpc_commit_bincode_write streams, tests/test_gpu_commit_serde.py)
    println!("{} commit_bincode_len {} commit_bincode_sha256 {}", name, dump.len(), hex(&Sha256::digest(&dump)));
    let fb = std::mem::size_of::<F>();
    let n_comm = u64::from_le_bytes(dump[0..8].try_into().unwrap()) as usize;
    let comm_bytes = &dump[8..8 + n_comm * fb];
    println!("{} comm_sha256 {}", name, hex(&Sha256::digest(comm_bytes)));
    let e1: F = bincode::deserialize(&comm_bytes[fb..2 * fb]).unwrap();
    println!("{} comm_row0_col1_repr {}", name, hex(e1.to_repr().as_ref()));
    let mut off = 8 + n_comm * fb;
    let n_coeffs = u64::from_le_bytes(dump[off..off + 8].try_into().unwrap()) as usize;
    off += 8 + n_coeffs * fb + 24;
    let n_hashes = u64::from_le_bytes(dump[off..off + 8].try_into().unwrap()) as usize;
    off += 8;
    let mut h = Sha256::new();
    for i in 0..n_hashes {
        let d = &dump[off + 8..off + 40];
        if i == 0 {
            println!("{} leaf0 {}", name, hex(d));
        }
        h.update(d);
        off += 40;
    }
    println!("{} hashes_sha256 {}", name, hex(&h.finalize()));
    if with_proof {
        // eval point 0x1234567; outer = (x^n_per_row)^r, as in tests/golden/make_golden.py
        let x = F::from(0x1234567u64);
        let inner = powers(x, np);
        let xr = *inner.last().unwrap() * x;
        let outer = powers(xr, nr);
        let mk = || {
            let mut tr = Transcript::new(b"test transcript");
            tr.append_message(b"polycommit", root.as_ref());
            tr.append_message(b"ncols", &(enc.get_n_col_opens() as u64).to_be_bytes()[..]);
            tr
        };
        let pf = comm.prove(&outer[..], enc, &mut mk()).unwrap();
        let bytes = bincode::serialize(&pf).unwrap();
        println!("{} proof_len {} proof_sha256 {} proof_blake3 {}", name, bytes.len(), hex(&Sha256::digest(&bytes)),
                 hex(blake3::hash(&bytes).as_bytes()));
        let ev = pf.verify(root.as_ref(), &outer[..], &inner[..], enc, &mut mk()).unwrap();
        println!("{} eval {}", name, hex_be(ev.to_repr().as_ref()));
    }
}

fn main() {
    case("ligero_ft63_2e10_iota", &coeffs::<Ft63>(1024, "iota", 0), &LigeroEncodingRho::<Ft63, U1, U2>::new(1024), true);
    case("ligero_ft63_1000_rand", &coeffs::<Ft63>(1000, "rand", 1), &LigeroEncodingRho::<Ft63, U1, U2>::new(1000), true);
    case("ligero_ft127_777_rho14", &coeffs::<Ft127>(777, "rand", 2), &LigeroEncodingRho::<Ft127, U1, U4>::new(777), true);
    case("ligero_ft191_dims_100_256", &coeffs::<Ft191>(950, "rand", 3), &LigeroEncodingRho::<Ft191, U38, U39>::new_from_dims(100, 256), true);
    case("ligero_ft255_2e12_iota", &coeffs::<Ft255>(4096, "iota", 0), &LigeroEncodingRho::<Ft255, U1, U2>::new(4096), true);
    case("ligero_ft255_3000_rand", &coeffs::<Ft255>(3000, "rand", 4), &LigeroEncodingRho::<Ft255, U1, U2>::new(3000), true);
    case("sdig_ft255_600_seed0", &coeffs::<Ft255>(600, "rand", 5), &SdigEncodingS::<Ft255, SdigCode3>::new(600, 0), true);
    case("sdig_ft63_900_code5", &coeffs::<Ft63>(900, "rand", 6), &SdigEncodingS::<Ft63, SdigCode5>::new(900, 77), true);
    case("sdig_ft127_2000_seed9", &coeffs::<Ft127>(2000, "iota", 0), &SdigEncodingS::<Ft127, SdigCode3>::new(2000, 9), false);
}
