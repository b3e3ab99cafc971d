// +build !hip

// boss_nohip.go -- companion of boss_hip.go for the default build of will-rowe/groot: the hook in mapReads compiles away.
package pipeline

const hipEnabled = false

func (theBoss *theBoss) mapReadsHIP() error { return nil }
