/* Stub of the all-gather zg_prover_set_shard asks for (tools/shard_compute_leg.py): this rank's partial sums go into slot 0,
 * rank 1's slot is filled with a fixed valid point (the generator, so that no commitment of a slice that holds none of a
 * column's non-zero rows comes out as the identity, which EvmTranscript refuses), every further slot with zeros -- the XYZZ
 * identity (zz = 0).  No communication: what remains is one rank's COMPUTE leg of a point-range-sharded proof.  The proofs
 * made this way are not valid and are not checked. */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct stub_user {
    uint64_t world;
    unsigned char filler[128]; /* one XYZZ point in the library's packed Montgomery form */
};

int zg_stub_exchange(void *user, const void *send, size_t nbytes, void *recv) {
    const struct stub_user *u = (const struct stub_user *)user;
    memset(recv, 0, nbytes * u->world);
    memcpy(recv, send, nbytes);
    if (u->world > 1)
        for (size_t off = 0; off + 128 <= nbytes; off += 128) memcpy((char *)recv + nbytes + off, u->filler, 128);
    return 0;
}
