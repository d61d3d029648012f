// tools/probe_stark_rings.rs -- prints the conventions of stark-rings @ 886a89f that no in-tree reference test pins
// (SURVEY.md App. C), in the form lf_set_ring_tables / the oracle's lfo_set_ring expect.
//
// NOT built here (no Rust toolchain in the image).  Usage on a machine with the reference workspace:
//   cp tools/probe_stark_rings.rs /path/to/latticefold/crates/cyclotomic-rings/examples/probe.rs
//   cargo run -p cyclotomic-rings --example probe > stark_rings_tables.json
use ark_ff::{Field, PrimeField, Zero, One};
use ark_serialize::{CanonicalSerialize, Compress};
use stark_rings::{
    balanced_decomposition::DecomposeToVec,
    cyclotomic_ring::{models::goldilocks::{Fq, Fq3, RqNTT, RqPoly}, CRT, ICRT},
    PolyRing,
};

fn c(x: Fq) -> u64 { x.into_bigint().0[0] }

fn main() {
    // 1. non-residue: Y^3 where Y = (0,1,0)
    let y = Fq3::new(Fq::zero(), Fq::one(), Fq::zero());
    let y3 = y * y * y;
    let bp: Vec<Fq> = y3.to_base_prime_field_elements().collect();
    println!("{{\"nonres\": {},", c(bp[0]));

    // 2. image of X in every slot: CRT of the monomial X
    let mut x = vec![Fq::zero(); 24];
    x[1] = Fq::one();
    let xn: RqNTT = RqPoly::from(x).crt();
    let slots: Vec<Vec<u64>> = xn.coeffs().iter().map(|s| s.to_base_prime_field_elements().map(c).collect()).collect();
    println!(" \"y\": {:?},", slots);

    // 3. full CRT matrix (unit monomials) as a cross-check of the (nonres, y) parametrisation
    let mut rows = vec![];
    for j in 0..24 {
        let mut e = vec![Fq::zero(); 24];
        e[j] = Fq::one();
        let n: RqNTT = RqPoly::from(e).crt();
        let w: Vec<u64> = n.coeffs().iter().flat_map(|s| s.to_base_prime_field_elements()).map(c).collect();
        rows.push(w);
    }
    println!(" \"crt_of_monomials\": {:?},", rows);

    // 4. balanced digits on edge coefficients, bases 2^16 (4 digits) and 2 (16 digits)
    let p: u128 = 18446744069414584321;
    let cases: Vec<u128> = vec![0, 1, p - 1, 32767, 32768, 32769, p - 32768, p - 32769, (p - 1) / 2, (p + 1) / 2, 65535, 65536];
    let mut out = vec![];
    for v in cases {
        let mut e = vec![Fq::zero(); 24];
        e[0] = Fq::from(v);
        let el = vec![RqPoly::from(e)];
        let d16: Vec<u64> = el.decompose_to_vec(1u128 << 16, 4)[0].iter().map(|r| c(r.coeffs()[0])).collect();
        let d2: Vec<u64> = el.decompose_to_vec(2u128, 16)[0].iter().map(|r| c(r.coeffs()[0])).collect();
        out.push((v as u64, d16, d2));
    }
    println!(" \"digit_cases\": {:?},", out);

    // 4b. structure constants of the extension fields in the crate's OWN coordinate basis: e_i * e_j for the unit vectors of
    //     to_base_prime_field_elements order.  Goldilocks Fq3 (tau = 3) and BabyBear Fq9 (tau = 9; a tower basis shows up here as a
    //     permuted binomial table).  lf_set_ext_basis wants T with ext = T * int; for a permutation it can be read off the table:
    //     find the generator g (a unit vector with g^tau in the base field) and list the unit vector each power g^k lands on.
    fn tensor<F: Field>(tau: usize, name: &str) {
        let unit = |i: usize| F::from_base_prime_field_elems(&(0..tau).map(|k| if k == i { F::BasePrimeField::one() } else { F::BasePrimeField::zero() }).collect::<Vec<_>>()).unwrap();
        let mut t = vec![];
        for i in 0..tau { for j in 0..tau {
            let pr = unit(i) * unit(j);
            let w: Vec<String> = pr.to_base_prime_field_elements().map(|x| x.into_bigint().to_string()).collect();
            t.push(w);
        } }
        println!(" \"ext_mul_tensor_{}\": {:?},", name, t);
    }
    tensor::<Fq3>(3, "goldilocks_fq3");
    {
        use stark_rings::cyclotomic_ring::models::babybear::{Fq9, RqNTT as BbNTT, RqPoly as BbPoly, Fq as BbFq};
        tensor::<Fq9>(9, "babybear_fq9");
        // BabyBear CRT of the 72 unit monomials (dense 72 x 72 matrix, columns = coefficient index)
        let mut rows = vec![];
        for j in 0..72 {
            let mut e = vec![BbFq::zero(); 72];
            e[j] = BbFq::one();
            let n: BbNTT = BbPoly::from(e).crt();
            let w: Vec<u64> = n.coeffs().iter().flat_map(|s| s.to_base_prime_field_elements()).map(|x| x.into_bigint().0[0]).collect();
            rows.push(w);
        }
        println!(" \"babybear_crt_of_monomials\": {:?},", rows);
    }

    // 6. LatticeFold+ conventions on the Frog ring (oracle/lfp.h restates them): exp, decompose_to_vec, gadget digits
    {
        use stark_rings::cyclotomic_ring::models::frog_ring::{Fq as FrogFq, RqPoly as FrogPoly};
        use stark_rings::{balanced_decomposition::GadgetDecompose, exp};
        use stark_rings_linalg::Matrix;
        let sgn = |v: i64| if v >= 0 { FrogFq::from(v as u64) } else { -FrogFq::from((-v) as u64) };
        let show = |r: &FrogPoly| r.coeffs().iter().map(|x| x.into_bigint().0[0]).collect::<Vec<u64>>();
        let exps: Vec<Vec<u64>> = [-7i64, -3, -1, 0, 1, 3, 7].iter().map(|a| show(&exp::<FrogPoly>(sgn(*a)).unwrap())).collect();
        println!(" \"frog_exp\": {:?},", exps);
        let vals: Vec<FrogFq> = [4i64, -4, 5, -5, 12, -12, 28, -28, 31, -31].iter().map(|v| sgn(*v)).collect();
        let digs: Vec<Vec<u64>> = vals.decompose_to_vec(8u128, 2).iter().map(|d| d.iter().map(|x| x.into_bigint().0[0]).collect()).collect();
        println!(" \"frog_digits_b8_k2\": {:?},", digs);
        let mut e = vec![FrogFq::zero(); 16];
        for (i, v) in e.iter_mut().enumerate() { *v = FrogFq::from(123456789u64 * (i as u64 + 1)); }
        let m: Matrix<FrogPoly> = vec![vec![FrogPoly::from(e), FrogPoly::from(vec![FrogFq::one(); 16])]].into();
        let g = m.gadget_decompose(8u128, 22);
        println!(" \"frog_gadget_row\": {:?},", g.vals[0].iter().map(|r| show(r)).collect::<Vec<_>>());
    }

    // 5. bytes of ONE serialized ring element (the per-element layout lf_wire.cpp assumes: 24 words x 8 bytes LE, no prefix)
    let mut e = vec![Fq::zero(); 24];
    for (i, v) in e.iter_mut().enumerate() { *v = Fq::from(1000u64 + i as u64); }
    let n: RqNTT = RqPoly::from(e).crt();
    let mut bytes = vec![];
    n.serialize_with_mode(&mut bytes, Compress::Yes).unwrap();
    let words: Vec<u64> = n.coeffs().iter().flat_map(|s| s.to_base_prime_field_elements()).map(c).collect();
    println!(" \"serialized_element\": {{\"len\": {}, \"bytes\": {:?}, \"flat_words\": {:?}}}}}", bytes.len(), bytes, words);
}
