// Package minaverify: cgo binding of libminaverify.so for the Go operator (INTEGRATION.md section 2).
// Written next to the header it binds; not built in this repository (no Go toolchain there).
package minaverify

/*
#cgo LDFLAGS: -lminaverify
#include <stdlib.h>
#include "mina_verify.h"
*/
import "C"

import (
	"errors"
	"unsafe"
)

type Ctx struct{ p *C.mina_ctx }

func lastError() error { return errors.New(C.GoString(C.mina_last_error())) }

func New(device int) (*Ctx, error) {
	var p *C.mina_ctx
	if rc := C.mina_ctx_create(C.int(device), &p); rc != 0 {
		return nil, lastError()
	}
	return &Ctx{p}, nil
}

func (c *Ctx) Close() { C.mina_ctx_destroy(c.p) }

func (c *Ctx) SrsCreate(curve int, depth uint32) error {
	if rc := C.mina_srs_create(c.p, C.int(curve), C.uint32_t(depth)); rc != 0 {
		return lastError()
	}
	return nil
}

// AccumulatorCheckMulti: one deterministic verdict per proof (len(sg)/64 proofs).
func (c *Ctx) AccumulatorCheckMulti(curve int, k uint32, pre, sg []byte) ([]bool, error) {
	n := len(sg) / 64
	out := make([]byte, n)
	rc := C.mina_accumulator_check_multi(c.p, C.int(curve), C.uint32_t(k), C.size_t(n),
		(*C.uint8_t)(unsafe.Pointer(&pre[0])), (*C.uint8_t)(unsafe.Pointer(&sg[0])), (*C.uint8_t)(unsafe.Pointer(&out[0])))
	if rc != 0 {
		return nil, lastError()
	}
	v := make([]bool, n)
	for i := range out {
		v[i] = out[i] != 0
	}
	return v, nil
}

// VerifyMinaState: drop-in for the operator's cgo call of verify_mina_state_ffi (bincode MinaStateProof + MinaStatePubInputs).
func VerifyMinaState(proof, pub []byte) bool {
	if len(proof) == 0 || len(pub) == 0 {
		return false
	}
	return bool(C.mina_verify_state((*C.uint8_t)(unsafe.Pointer(&proof[0])), C.size_t(len(proof)), (*C.uint8_t)(unsafe.Pointer(&pub[0])), C.size_t(len(pub))))
}

// VerifyAccountInclusionFFI: drop-in for verify_account_inclusion_ffi (bincode MinaAccountProof + MinaAccountPubInputs).
func VerifyAccountInclusionFFI(proof, pub []byte) bool {
	if len(proof) == 0 || len(pub) == 0 {
		return false
	}
	return bool(C.mina_verify_account((*C.uint8_t)(unsafe.Pointer(&proof[0])), C.size_t(len(proof)), (*C.uint8_t)(unsafe.Pointer(&pub[0])), C.size_t(len(pub))))
}

// VerifyAccountInclusion: the Merkle part of verify_account_inclusion_ffi on the reference's byte contract.
func (c *Ctx) VerifyAccountInclusion(proof, pub, leafHash []byte) (bool, error) {
	var verdict C.uint8_t
	// the C side takes arrays of pointers: pin the two buffers in C memory for the call (cgo pointer rules)
	cp, cq := C.CBytes(proof), C.CBytes(pub)
	defer C.free(cp)
	defer C.free(cq)
	p, q := (*C.uint8_t)(cp), (*C.uint8_t)(cq)
	pl, ql := C.size_t(len(proof)), C.size_t(len(pub))
	rc := C.mina_verify_account_inclusion(c.p, 1, &p, &pl, &q, &ql, (*C.uint8_t)(unsafe.Pointer(&leafHash[0])), &verdict)
	if rc != 0 {
		return false, lastError()
	}
	return verdict == 1, nil
}
