// oracle/repin/src/main.rs -- prints, from the REAL conroi/lcpc crates, the values that this repository's golden
// fixtures (tests/golden/commit_cases.json, made by the Python restatement) claim for the same inputs.
// Equal output  =>  every third-party convention marked [3P] in oracle/ is pinned at once:
//   ff_derive Montgomery form / to_repr / From<u64>, fffft root + output order, BLAKE3 leaf format, Merkle layout,
//   merlin transcript, ChaCha20 + Field::random, rand Uniform, rand_core seed_from_u64 (Brakedown matgen), bincode.
// Not compiled in this repo's image (no Rust toolchain); written against the API in /root/reference:
//   LcCommit::commit / prove / get_root (lcpc-2d/src/lib.rs:270-312), LigeroEncoding::new (ligero lib.rs:121-124),
//   SdigEncoding::new (brakedown lib.rs:103-110), test transcript set-up (ligero tests.rs:243-245).
use blake3::Hasher as Blake3;
use ff::{Field, PrimeField};
use lcpc_2d::{LcCommit, LcEncoding};
use lcpc_brakedown_pc::SdigEncoding;
use lcpc_ligero_pc::LigeroEncoding;
use lcpc_test_fields::{ft127::Ft127, ft255::Ft255, ft63::Ft63};
use merlin::Transcript;

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
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

fn case<F, E>(name: &str, coeffs: &[F], enc: &E, with_proof: bool)
where
    F: PrimeField + serde::Serialize,
    E: LcEncoding<F = F>,
{
    let comm = LcCommit::<Blake3, E>::commit(coeffs, enc).unwrap();
    let root = comm.get_root();
    println!("{} dims {} {} {}", name, comm.get_n_rows(), comm.get_n_per_row(), comm.get_n_cols());
    println!("{} root {}", name, hex(root.as_ref()));
    if with_proof {
        // eval point 0x1234567; outer = (x^n_per_row)^r, as in tests/golden/make_golden.py
        let x = F::from(0x1234567u64);
        let inner = powers(x, comm.get_n_per_row());
        let xr = *inner.last().unwrap() * x;
        let outer = powers(xr, comm.get_n_rows());
        let mut tr = Transcript::new(b"test transcript");
        tr.append_message(b"polycommit", root.as_ref());
        tr.append_message(b"ncols", &(enc.get_n_col_opens() as u64).to_be_bytes()[..]);
        let pf = comm.prove(&outer[..], enc, &mut tr).unwrap();
        let bytes = bincode::serialize(&pf).unwrap();
        println!("{} proof_len {} proof_blake3 {}", name, bytes.len(), hex(blake3::hash(&bytes).as_bytes()));
    }
}

fn main() {
    // golden "ligero_ft63_2e10_iota": c_i = i + 1, i < 1024
    let c63: Vec<Ft63> = (1..=1024u64).map(Ft63::from).collect();
    case("ligero_ft63_2e10_iota", &c63, &LigeroEncoding::<Ft63>::new(c63.len()), true);
    // golden "ligero_ft255_2e12_iota": c_i = i + 1, i < 4096
    let c255: Vec<Ft255> = (1..=4096u64).map(Ft255::from).collect();
    case("ligero_ft255_2e12_iota", &c255, &LigeroEncoding::<Ft255>::new(c255.len()), true);
    // golden "sdig_ft127_2000_seed9": c_i = i + 1, i < 2000, SdigCode3, matgen seed 9 (commit only)
    let c127: Vec<Ft127> = (1..=2000u64).map(Ft127::from).collect();
    case("sdig_ft127_2000_seed9", &c127, &SdigEncoding::<Ft127>::new(c127.len(), 9), false);
}
